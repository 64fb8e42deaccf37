"""Host side of the face detector and its oracle (no GPU): the product's layer list equals the
oracle's, the checkpoint converter folds batch norms correctly, and the oracle's decode / suppression
/ letterbox answer hand-worked cases."""
import numpy as np
import pytest


def test_layer_lists_agree():
    from frhot import detector as D
    from oracle import yolo_face as O
    assert [(n, k, ci, co) for n, k, _s, ci, co, _a in O.conv_specs()] == D.CONVS
    assert len(D.CONVS) == 72
    assert all(len(n) + 2 <= 15 for n, *_ in D.CONVS), "blob names hold 15 characters"
    assert D.MACS_PER_FRAME == 4_165_420_800
    blob = D.pack_blob(D.synthetic_weights(1))
    assert bytes(blob[:8]) == b"FRHOTW01"


def test_converter_folds_batch_norm():
    """A made-up state dict with the checkpoint's key names: conv -> batch norm (eval) equals the folded
    convolution + bias, kernels come out as [cout, k, k, cin], and the heads' last layers keep their bias."""
    import torch
    import torch.nn.functional as F
    from frhot import detector as D
    rng = np.random.default_rng(0)
    sd = {}

    def put(prefix, k, cin, cout, bn=True):
        sd[prefix + (".conv.weight" if bn else ".weight")] = rng.standard_normal((cout, cin, k, k)).astype(np.float32)
        if bn:
            sd[prefix + ".bn.weight"] = rng.uniform(0.5, 1.5, cout).astype(np.float32)
            sd[prefix + ".bn.bias"] = rng.standard_normal(cout).astype(np.float32)
            sd[prefix + ".bn.running_mean"] = rng.standard_normal(cout).astype(np.float32)
            sd[prefix + ".bn.running_var"] = rng.uniform(0.5, 2.0, cout).astype(np.float32)
        else:
            sd[prefix + ".bias"] = rng.standard_normal(cout).astype(np.float32)

    names = {}
    for name, k, cin, cout in D.CONVS:
        p = name.split(".")
        if p[0] == "h":
            mod = {"box": "cv2", "cls": "cv3", "kpt": "cv4"}[p[1][:3]]
            prefix = f"model.22.{mod}.{p[1][3]}.{p[2]}"
            put(prefix, k, cin, cout, bn=p[2] != "2")
        else:
            i = int(p[0][1:])
            prefix = f"model.{i}" + ("" if len(p) == 1 else (f".m.{p[1][1:]}.{p[2]}" if len(p) == 3 else f".{p[1]}"))
            put(prefix, k, cin, cout)
        names[name] = prefix
    out = D.from_ultralytics_state_dict(sd)
    for name, k, cin, cout in (D.CONVS[0], D.CONVS[3], D.CONVS[5], D.CONVS[-1], D.CONVS[-3]):
        prefix = names[name]
        x = torch.from_numpy(rng.standard_normal((1, cin, 9, 9)).astype(np.float32))
        if prefix + ".conv.weight" in sd:
            y = F.conv2d(x, torch.from_numpy(sd[prefix + ".conv.weight"]), padding=k // 2)
            y = F.batch_norm(y, torch.from_numpy(sd[prefix + ".bn.running_mean"]), torch.from_numpy(sd[prefix + ".bn.running_var"]),
                             torch.from_numpy(sd[prefix + ".bn.weight"]), torch.from_numpy(sd[prefix + ".bn.bias"]), False, 0.0, 1e-3)
        else:
            y = F.conv2d(x, torch.from_numpy(sd[prefix + ".weight"]), torch.from_numpy(sd[prefix + ".bias"]), padding=k // 2)
        w = torch.from_numpy(out[name + ".w"]).permute(0, 3, 1, 2)
        z = F.conv2d(x, w, torch.from_numpy(out[name + ".b"]), padding=k // 2)
        assert out[name + ".w"].shape == (cout, k, k, cin)
        assert torch.allclose(y, z, atol=2e-4, rtol=1e-4), name


def test_oracle_letterbox_geometry():
    from oracle import yolo_face as O
    img = np.full((480, 640, 3), (10, 20, 30), np.uint8)          # BGR
    x, r, (left, top) = O.letterbox(img, auto=False)              # the full square
    assert x.shape == (640, 640, 3) and r == 1.0 and (left, top) == (0, 80)
    assert np.allclose(x[79, 0], 114 / 255) and np.allclose(x[80, 0], [30 / 255, 20 / 255, 10 / 255])      # RGB
    assert np.allclose(x[559, 639], [30 / 255, 20 / 255, 10 / 255]) and np.allclose(x[560, 0], 114 / 255)
    x, r, (left, top) = O.letterbox(img)                          # a frame on its own: the minimal rectangle
    assert x.shape == (480, 640, 3) and (left, top) == (0, 0) and np.allclose(x[0, 0], [30 / 255, 20 / 255, 10 / 255])
    x, r, (left, top) = O.letterbox(np.zeros((469, 648, 3), np.uint8))
    assert x.shape == (480, 640, 3) and (left, top) == (0, 8) and x[7, 0, 0] == np.float32(114 / 255) and x[8, 0, 0] == 0
    assert x[470, 0, 0] == 0 and x[471, 0, 0] == np.float32(114 / 255)          # 463 rows, 8 above and 9 below
    tall = np.zeros((1280, 320, 3), np.uint8)
    x, r, (left, top) = O.letterbox(tall)
    assert x.shape == (640, 160, 3) and r == 0.5 and (left, top) == (0, 0)
    x, r, (left, top) = O.letterbox(tall, auto=False)
    assert r == 0.5 and (left, top) == (240, 0) and x[0, 239, 0] == np.float32(114 / 255) and x[0, 240, 0] == 0
    small = np.zeros((100, 50, 3), np.uint8)                        # enlarged
    x, r, (left, top) = O.letterbox(small)
    assert x.shape == (640, 320, 3) and r == 6.4 and (left, top) == (0, 0)
    for h, w in ((333, 517), (719, 1279), (64, 640), (641, 640), (1, 1), (5, 4000)):
        x, _r, _p = O.letterbox(np.zeros((h, w, 3), np.uint8))
        assert x.shape[0] % 32 == 0 and x.shape[1] % 32 == 0 and 32 <= x.shape[0] <= 640 and 32 <= x.shape[1] <= 640


def test_oracle_decode_hand_case():
    """One anchor with a known answer: level 1 (stride 16), cell (row 3, column 5); every side's
    bins put all the mass on one bin, so the distances are that bin's number."""
    from oracle import yolo_face as O
    outs = []
    for l, hw in enumerate((80, 40, 20)):
        box = np.zeros((1, hw, hw, 64), np.float32)
        cls = np.full((1, hw, hw, 1), -20.0, np.float32)
        kpt = np.zeros((1, hw, hw, 15), np.float32)
        outs.append((box, cls, kpt))
    box, cls, kpt = outs[1]
    for side, b in enumerate((2, 3, 4, 5)):
        box[0, 3, 5, side * 16 + b] = 60.0
    cls[0, 3, 5, 0] = 2.0
    kpt[0, 3, 5, 0:3] = (0.25, -0.5, 0.0)
    (boxes, scores, kpts), = O.decode(outs)
    assert len(scores) == 1 and abs(scores[0] - 1 / (1 + np.exp(-2.0))) < 1e-6
    # anchor centre (5.5, 3.5); x1 = 3.5, y1 = 0.5, x2 = 9.5, y2 = 8.5 cells -> times 16
    assert np.allclose(boxes[0], [6.5 * 16, 4.5 * 16, 6 * 16, 8 * 16], atol=1e-3)
    assert np.allclose(kpts[0, 0], [(0.5 + 5) * 16, (-1.0 + 3) * 16, 0.5], atol=1e-5)


def test_oracle_suppression():
    from oracle import yolo_face as O
    boxes = np.array([[100, 100, 50, 50], [102, 100, 50, 50], [300, 300, 40, 40], [100, 100, 50, 50]], np.float32)
    scores = np.array([0.9, 0.95, 0.5, 0.9], np.float32)
    assert O.nms(boxes, scores, iou=0.7) == [1, 2]          # 0 and 3 overlap 1 by more than 0.7
    assert O.nms(boxes, scores, iou=1.0) == [1, 0, 3, 2]    # nothing suppressed: score order, ties by index
    assert O.nms(boxes, scores, iou=0.7, max_det=1) == [1]


def test_region_of_matches_oracle_rule():
    from frhot.detector import region_of
    row = np.zeros(20, np.float32)
    row[:5] = (10.6, 20.2, 110.9, 140.7, 0.875)
    row[5:11] = (30.9, 50.1, 0.9, 80.2, 51.7, 0.8)
    r = region_of(row)
    assert r == {"x": 10, "y": 20, "w": 100, "h": 120, "right_eye": (30, 50), "left_eye": (80, 51), "confidence": 0.875}


def test_detect_batcher_groups_frames_and_isolates_failures():
    """Concurrent single-frame requests share device calls; a frame the detector refuses fails its own
    request only; extract_faces needs an anti-spoofing policy."""
    import threading
    from frhot.batcher import DetectBatcher

    class FakeDetector:
        max_batch = 8

        def __init__(self):
            self.calls = []

        def detect(self, images, conf=0.25):
            self.calls.append(len(images))
            if any(im.shape[0] == 13 for im in images):
                raise RuntimeError("refused")
            out = []
            for im in images:
                row = np.zeros((1, 20), np.float32)
                row[0, :5] = (1, 2, 1 + im.shape[1] // 2, 2 + im.shape[0] // 2, 0.9)
                out.append(row)
            return out

    det = FakeDetector()
    b = DetectBatcher(det, max_delay_s=0.2)
    try:
        frames = [np.zeros((20 + i, 40, 3), np.uint8) for i in range(6)] + [np.zeros((13, 40, 3), np.uint8)]
        results = [None] * len(frames)

        def run(i):
            try:
                results[i] = b.detect(frames[i])
            except Exception as e:      # noqa: BLE001
                results[i] = e
        threads = [threading.Thread(target=run, args=(i,)) for i in range(len(frames))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert isinstance(results[6], RuntimeError)
        for i in range(6):
            assert results[i].shape == (1, 20) and results[i][0, 3] == 2 + (20 + i) // 2
        assert det.calls[0] > 1, "the first device call carried several frames"
        with pytest.raises(ValueError):
            b.detect(np.zeros((4, 4), np.uint8))
        with pytest.raises(ValueError):
            b.extract_faces(frames[0])
        faces = b.extract_faces(frames[0], expand_percentage=10, assume_real=True)
        assert faces[0]["is_real"] is True and faces[0]["facial_area"]["w"] == 20 + 2 and faces[0]["confidence"] == 0.9
        faces = b.extract_faces(frames[0], anti_spoof=lambda crop: False)
        assert faces[0]["is_real"] is False
    finally:
        b.close()
