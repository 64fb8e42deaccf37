"""The face detector on the GPU (C ABI: frhot_det_*) against oracle/yolo_face.py: letterbox bit for
bit, every block's activations, the decoded candidates, and the regions after suppression.
PARITY UNPINNED: the oracle restates ultralytics / deepface from their published behaviour."""
import numpy as np
import pytest

from det_util import calibrated_weights, frames

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def det(gpu_ctx):
    from frhot import detector as D
    w, wq = calibrated_weights(seed=3, cls_bias=-6.0)
    d = D.Detector(gpu_ctx, D.pack_blob(w), max_batch=4)
    yield d, wq
    d.close()


def test_letterbox_bit_exact(det):
    from oracle import yolo_face as O
    d, _wq = det
    rng = np.random.default_rng(1)
    shapes = [(480, 640), (640, 640), (1280, 320), (100, 50), (333, 517), (719, 1279), (641, 640), (64, 640), (469, 648)]
    imgs = [(rng.random((h, w, 3)) * 255).astype(np.uint8) for h, w in shapes]
    big = (rng.random((500, 700, 3)) * 255).astype(np.uint8)
    imgs.append(big[20:420, 100:650])                     # a row-strided view
    for i, im in enumerate(imgs):
        want = O.letterbox(im)[0].astype(np.float16)      # the minimal rectangle of a frame on its own
        d.detect([im])
        got = d.debug_read("in", 1, want.shape[0], want.shape[1], 4)
        assert np.array_equal(got[0, :, :, :3], want), f"image {i} {im.shape}"
        assert not got[0, :, :, 3].any()


def test_blocks_match_oracle(det):
    from oracle import yolo_face as O
    d, wq = det
    where = {"m0": ("m0", 2, 16, 0), "m1": ("m1", 4, 32, 0), "m2": ("m2", 4, 32, 0), "m3": ("m3", 8, 64, 0),
             "m4": ("cat14", 8, 192, 128), "m5": ("m5", 16, 128, 0), "m6": ("cat11", 16, 384, 256), "m7": ("m7", 32, 256, 0),
             "m8": ("m8", 32, 256, 0), "m9": ("cat20", 32, 384, 128), "m12": ("cat17", 16, 192, 64), "m15": ("o3", 8, 64, 0),
             "m18": ("o4", 16, 128, 0), "m21": ("o5", 32, 256, 0)}
    same = frames(seed=9, n=1, lo=500, hi=640)[0]
    for imgs in ([same, same[::-1].copy()], [frames(seed=10, n=1, lo=240, hi=420)[0]]):       # a batch of two, a small frame
        d.detect(imgs)
        x = np.stack([O.letterbox(im)[0] for im in imgs])
        H, W = x.shape[1:3]
        taps = {}
        outs = O.forward(x, wq, taps)
        for name, (tensor, div, c, c0) in where.items():
            ref = taps[name]
            got = d.debug_read(tensor, len(imgs), H // div, W // div, c)[..., c0:c0 + ref.shape[-1]].astype(np.float32)
            err, scale = np.abs(got - ref).max(), np.abs(ref).max()
            assert err <= 1e-2 * scale, f"{name}: max err {err} vs scale {scale}"
        for l, (box, cls, kpt) in enumerate(outs):
            h, w = box.shape[1:3]
            for nm, ref, c in (("box", box, 64), ("cls", cls, 16), ("kpt", kpt, 16)):
                got = d.debug_read(f"{nm}{l}", len(imgs), h, w, c)[..., :ref.shape[-1]].astype(np.float32)
                err, scale = np.abs(got - ref).max(), np.abs(ref).max()
                assert err <= 1.5e-2 * scale, f"{nm}{l}: max err {err} vs scale {scale}"


def _match(got, ref_boxes, ref_scores, ref_kpts, tol_px, tol_s):
    """every reference candidate well clear of the threshold has a device row with the same numbers"""
    used = set()
    for b, s, k in zip(ref_boxes, ref_scores, ref_kpts):
        if abs(s - 0.25) < 0.03:
            continue
        best = None
        for j, row in enumerate(got):
            if j in used:
                continue
            if abs(row[4] - s) <= tol_s and np.abs(row[:4] - b).max() <= tol_px:
                best = j
                break
        assert best is not None, f"no device row for box {b} score {s}"
        used.add(best)
        assert np.abs(got[best][5:].reshape(5, 3)[:, :2] - k[:, :2]).max() <= tol_px
        assert np.abs(got[best][5:].reshape(5, 3)[:, 2] - k[:, 2]).max() <= tol_s
    return len(used)


def test_candidates_match_oracle(det):
    """iou = 1 suppresses nothing: the device's rows are its decoded candidates (mapped back to the
    frame), compared with the oracle's decode of its own float32 head maps."""
    from oracle import yolo_face as O
    d, wq = det
    imgs = frames(seed=11, n=3)
    got = d.detect(imgs, conf=0.25, iou=1.0, max_faces=300)
    n_checked = 0
    for im, rows in zip(imgs, got):
        x, r, (left, top) = O.letterbox(im)
        boxes, scores, kpts = O.decode(O.forward(x[None], wq))[0]
        H, W = im.shape[:2]
        xyxy = np.stack([boxes[:, 0] - boxes[:, 2] / 2, boxes[:, 1] - boxes[:, 3] / 2, boxes[:, 0] + boxes[:, 2] / 2,
                         boxes[:, 1] + boxes[:, 3] / 2], 1)
        xyxy[:, [0, 2]] = np.clip((xyxy[:, [0, 2]] - left) / r, 0, W)
        xyxy[:, [1, 3]] = np.clip((xyxy[:, [1, 3]] - top) / r, 0, H)
        kp = kpts.copy()
        kp[:, :, 0] = np.clip((kp[:, :, 0] - left) / r, 0, W)
        kp[:, :, 1] = np.clip((kp[:, :, 1] - top) / r, 0, H)
        assert abs(len(rows) - min(len(scores), 300)) <= max(2, len(scores) // 20)
        assert (np.diff(rows[:, 4]) <= 0).all()
        if len(scores) <= 300:
            n_checked += _match(rows, xyxy, scores, kp, tol_px=1.5 / min(r, 1.0), tol_s=0.02)
    assert n_checked >= 5


def test_regions_match_oracle(det):
    from oracle import yolo_face as O
    d, wq = det
    imgs = frames(seed=13, n=4)
    agree = total = 0
    for im in imgs:
        ref = O.regions(im, wq)
        got = d.regions(im)
        total += len(ref)
        for a in ref:
            for b in got:
                if all(abs(a[k] - b[k]) <= 2 for k in ("x", "y", "w", "h")) and abs(a["confidence"] - b["confidence"]) <= 0.02 \
                        and all(abs(p - q) <= 2 for p, q in zip(a["left_eye"] + a["right_eye"], b["left_eye"] + b["right_eye"])):
                    agree += 1
                    break
        assert abs(len(ref) - len(got)) <= max(1, len(ref) // 10)
    assert total >= 4 and agree >= 0.9 * total, f"{agree} of {total} regions agree"


def test_extract_faces_entries_and_errors(det, gpu_ctx):
    from frhot import FrhotError
    d, _wq = det
    im = frames(seed=13, n=1)[0]
    faces = d.extract_faces(im, expand_percentage=10)
    for f in faces:
        a = f["facial_area"]
        assert f["is_real"] is None and 0 <= a["x"] and a["x"] + a["w"] <= im.shape[1] and a["y"] + a["h"] <= im.shape[0]
        assert f["face"].shape[:2] == (a["h"], a["w"])
    with pytest.raises(FrhotError):
        d.detect([im] * 2, conf=1.5)
    with pytest.raises(FrhotError):
        d.debug_read("nope", 1, 1, 1, 16)
    assert d.detect([im] * 5)[4].shape == d.detect([im])[0].shape          # more frames than max_batch: several calls


def test_frame_to_identity_through_the_detector(det, gpu_ctx):
    """A frame goes through the GPU detector, the most confident region is cropped, embedded and
    searched on the same context: the self-contained path of the reference's recognise request
    (face_service.py:145-150,221-233 with legacy.detect_face in front).  Enrolling the crop of one
    frame makes that frame's face its own best match."""
    from frhot import quality
    from frhot.engine import Gallery
    d, _wq = det
    with pytest.raises(ValueError):
        d.as_extract_faces()
    quality.install_models(extract_faces=d.as_extract_faces(expand_percentage=10, assume_real=True))
    try:
        gal = Gallery(gpu_ctx, 4096, 64)
        crops = []
        for i, im in enumerate(frames(seed=21, n=3, lo=400, hi=700)):
            boxes, scores, distances = quality.detect_face(im)
            assert len(boxes) >= 1 and len(boxes) == len(scores) == len(distances)
            assert (np.diff(scores) <= 0).all() and (scores >= quality.CONF_THRESHOLD).all()
            x, y, w, h = boxes[0]
            assert w > 0 and h > 0
            crops.append(np.ascontiguousarray(im[y:y + h, x:x + w]))
        emb = gpu_ctx.embed(crops, gamma=True)
        gal.upsert(emb, 0, np.arange(3))
        s, r, _ = gpu_ctx.recognise(gal, crops, None, 1, gamma=True)
        assert r[:, 0].tolist() == [0, 1, 2] and (s[:, 0] > 0.999).all()
        gal.close()
    finally:
        quality.install_models()


def test_detect_and_recognise_from_two_threads(det, gpu_ctx):
    """Detector calls and fused recognitions arrive from different threads on one context (the
    reference's handlers run in a thread pool, face_service.py:43,231-233): both go through the
    context's two staging slots, and every answer equals the one the same call gives alone."""
    import threading
    from frhot import synth
    from frhot.engine import Gallery
    d, _wq = det
    fr = frames(seed=31, n=3)
    crops = synth.crops(seed=32, n=12, lo=120, hi=300)
    gal = Gallery(gpu_ctx, 4096, 64)
    gal.upsert(gpu_ctx.embed(crops), 0, np.arange(12))
    want_det = d.detect(fr)
    want_s, want_r, _ = gpu_ctx.recognise(gal, crops, None, 3)
    errors = []

    def detect_loop():
        try:
            for _ in range(6):
                got = d.detect(fr)
                assert all(np.array_equal(a, b) for a, b in zip(got, want_det))
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    def recognise_loop():
        try:
            for _ in range(12):
                s, r, _ = gpu_ctx.recognise(gal, crops, None, 3)
                assert np.array_equal(r, want_r) and np.array_equal(s, want_s)
        except Exception as e:      # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=detect_loop), threading.Thread(target=recognise_loop), threading.Thread(target=recognise_loop)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    gal.close()
    assert not errors, errors


def test_detect_batcher_on_the_gpu(det):
    """Eight requests in flight share device calls and each gets the rows a lone call gives."""
    from concurrent.futures import ThreadPoolExecutor
    from frhot.batcher import DetectBatcher
    d, _wq = det
    fr = frames(seed=41, n=8)
    want = [d.detect([f])[0] for f in fr]
    b = DetectBatcher(d, max_delay_s=0.05)
    try:
        with ThreadPoolExecutor(max_workers=8) as pool:
            got = list(pool.map(b.detect, fr))
        assert b.batches < len(fr) and b.items == len(fr)
        for g, w in zip(got, want):
            assert np.array_equal(g, w)
    finally:
        b.close()


def test_create_refuses_a_blob_of_another_network(gpu_ctx, weights_f32):
    from frhot import FrhotError, detector as D
    from frhot.weights import pack_blob
    with pytest.raises(FrhotError) as e:
        D.Detector(gpu_ctx, pack_blob(weights_f32), max_batch=1)
    assert e.value.code == -1 and "m0.w" in str(e.value)
    with pytest.raises(FrhotError):
        D.Detector(gpu_ctx, D.pack_blob(D.synthetic_weights(1)), max_batch=65)


def test_app_serves_frames_through_the_gpu_detector(det, gpu_ctx):
    """create_app with the detector registered: an employee's check-in request carries a whole frame,
    the service detects on the GPU (frames of requests in flight share device calls), crops the nearest
    face, and recognises it: enrolling a frame makes the same frame answer with the enrolled id."""
    import asyncio
    import base64
    import io

    import httpx
    from PIL import Image

    from frhot import quality
    from frhot.api import create_app
    d, _wq = det
    with pytest.raises(ValueError):
        create_app(gpu_ctx, detector=d, quality_gates=True)                 # no anti-spoofing policy
    with pytest.raises(ValueError):
        create_app(gpu_ctx, detector=d, assume_real=True)                   # the gates are off
    app = create_app(gpu_ctx, capacity=256, api_key="k", max_delay_s=0.01, quality_gates=True, detector=d, assume_real=True)

    def b64(img):
        buf = io.BytesIO()
        Image.fromarray(img[:, :, ::-1]).save(buf, format="PNG")
        return base64.b64encode(buf.getvalue()).decode()
    # frames whose nearest detected face passes the distance gate (20..70) and the blur gate
    fr = []
    rng = np.random.default_rng(52)
    for im in frames(seed=51, n=24, lo=400, hi=640):
        im = np.clip(im.astype(np.float32) + rng.normal(0, 25, im.shape), 0, 255).astype(np.uint8)      # sharp enough for the blur gate
        boxes, scores, distances = quality.detect_face(im)
        if len(boxes):
            x, y, w, h = boxes[int(np.argmin(distances))]
            dist = quality.distance_face_to_camera((x, y, x + w, y + h), im.shape[1])
            if 20 <= dist <= 70 and w > 8 and h > 8 and quality.laplacian_variance_of(im[y:y + h, x:x + w]) >= quality.BLUR_THRESHOLD:
                fr.append(im)
    assert len(fr) >= 3, "the synthetic frames give too few usable detections"
    fr = fr[:6]

    async def run():
        H = {"X-API-Key": "k"}
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://frhot", timeout=120) as c:
            for i, im in enumerate(fr):
                r = await c.post("/create_face_img_base64", headers=H, json={
                    "img_base64": b64(im), "id": str(i), "name": f"P{i}", "role": "1", "store_id": "S", "is_update": True})
                assert r.status_code == 201, r.text
            rs = await asyncio.gather(*[c.post("/face_recog_img_base64", headers=H, json={
                "img_base64": b64(im), "role": "1", "store_id": "S"}) for im in fr])
            return [(r.status_code, r.json()) for r in rs]
    try:
        answers = asyncio.run(run())
        for i, (code, body) in enumerate(answers):
            assert code == 200 and body.get("id") == str(i), (i, code, body)
        assert app.state.det_batcher.items >= 2 * len(fr)
    finally:
        app.state.close()
    assert quality.models_installed()["extract_faces"] is False
