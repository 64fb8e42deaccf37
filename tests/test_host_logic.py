"""Host-side logic without a GPU: vote, gallery service bookkeeping, client reshaping,
the two handlers' status table, weight blob, synthetic data, micro-batcher."""
import asyncio
import base64
import io
import itertools
import struct
import threading

import numpy as np
import pytest

from fakes import HashEmbedder, NumpyGallery
from oracle import search as osr


# ------------------------------------------------------------------ vote ---
def test_vote_equals_reference_restatement_exhaustively():
    """Product vote == literal restatement of app.py:340-384 over every id pattern and a grid of scores."""
    from frhot.vote import vote
    score_sets = [
        (0.9, 0.8, 0.7, 0.6, 0.55), (0.9, 0.8, 0.7, 0.5, 0.4), (0.6, 0.58, 0.57, 0.2, 0.1),
        (0.56, 0.55, 0.53, 0.52, 0.5), (0.53, 0.5, 0.4, 0.3, 0.2), (0.54, 0.54, 0.54, 0.54, 0.54),
        (0.9, 0.9, 0.6, 0.6, 0.1),
    ]
    n = 0
    for scores in score_sets:
        for length in range(0, 6):
            for ids in itertools.product("ABC", repeat=length):
                hits = [(scores[i], {"id": ids[i], "name": ids[i], "slot": i}) for i in range(length)]
                assert vote(hits) == osr.vote(hits), (scores, ids)
                n += 1
    assert n > 2000


# ------------------------------------------------------- gallery service ---
def _service(dim=16, capacity=64):
    from frhot.gallery_service import GalleryService
    return GalleryService(None, capacity=capacity, dim=dim, gallery=NumpyGallery(dim, capacity))


def test_gallery_service_matches_oracle_collections_on_random_traffic():
    rng = np.random.default_rng(0)
    dim = 16
    svc, ref = _service(dim, 256), osr.Collections(dim=dim)
    cols = ["S1_Employees", "S1_Customers", "S2_Employees"]
    for c in cols:
        assert svc.create_collection(c) == ref.create_collection(c)
    assert svc.create_collection(cols[0]) == ref.create_collection(cols[0]) == (200, {"message": "Collection existed"})
    assert svc.create_collection(None) == ref.create_collection(None)
    base = rng.normal(size=(6, dim)).astype(np.float32)
    t = 0
    for step in range(400):
        op = rng.integers(0, 10)
        col = cols[rng.integers(0, 3)]
        store = col.split("_")[0] if rng.random() < 0.9 else "OTHER"
        pid = "ABCDEF"[rng.integers(0, 6)]
        vec = base["ABCDEF".index(pid)] + 0.25 * rng.normal(size=dim).astype(np.float32)
        if op < 5:
            t += 1
            ts = f"2025/01/01 00:{t // 60:02d}:{t % 60:02d}"
            a = svc.insert_point(col, vec, pid, "n" + pid, store, time_created=ts)
            b = ref.insert_point(col, vec, pid, "n" + pid, store, ts)
            assert a == b
        elif op < 6:
            assert svc.delete_point(col, pid) == ref.delete_point(col, pid)
        else:
            a = svc.search_point(col, vec, store)
            b = ref.search_point(col, vec, store)
            assert a[0] == b[0] and a[1]["message"] == b[1]["message"], (step, a, b)
            if a[1]["data"]:
                (sa, pa), (sb, pb) = a[1]["data"][0], b[1]["data"][0]
                assert pa == pb and abs(sa - sb) < 1e-5
        for c in cols:
            assert svc.point_count(c) == len(ref.cols[c])
    # nobody holds more than 10 faces per id in a collection
    for c in cols:
        for pid in "ABCDEF":
            assert sum(1 for p in ref.cols[c] if p[2]["id"] == pid) <= 10


def test_gallery_service_validation_and_eviction_order():
    svc = _service(8, 64)
    v = np.ones(8, np.float32)
    assert svc.insert_point("", v, "A", "a", "S")[0] == 404
    assert svc.insert_point("nope", v, "A", "a", "S") == (404, {"message": "Collection name not found or invalid!"})
    svc.create_collection("S_Employees")
    assert svc.insert_point("S_Employees", np.ones(7), "A", "a", "S") == (404, {"message": "Embedding not found or invalid!"})
    assert svc.insert_point("S_Employees", v, "", "a", "S") == (404, {"message": "ID not found or invalid!"})
    assert svc.insert_point("S_Employees", v, "A", "a", "S", is_update_id=True) == (404, {"message": "ID not found!"})
    for i in range(13):
        assert svc.insert_point("S_Employees", v, "A", "a", "S", time_created=f"t{i:02d}") == (201, {"message": "Point inserted"})
    times = sorted(p["payload"]["time_created"] for p in svc._points.values())
    assert times == [f"t{i:02d}" for i in range(3, 13)]          # oldest evicted, 10 remain
    assert svc.search_point("S_Employees", np.ones(3), "S")[0] == 404
    assert svc.search_point("", v, "S")[0] == 404
    assert svc.search_point("S_Employees", v, "ELSEWHERE") == (200, {"message": "Point not found", "data": []})
    assert svc.delete_point("S_Employees", "A") == (200, {"message": "Point deleted"})
    assert svc.point_count("S_Employees") == 0 and svc._rows.free_count() == svc.capacity
    assert svc.delete_point("S_Employees", None)[0] == 404
    # freed rows are reused
    svc.insert_point("S_Employees", v, "B", "b", "S")
    assert max(svc._points) < 13


def _same_answers(a, b, tol=2e-3):
    """Search answers equal up to the score rounding of the stand-in gallery (it keeps float32
    rows but snapshots carry fp16; the device gallery is fp16 throughout and round-trips exactly,
    which tests/test_gpu_hotpath.py asserts)."""
    assert len(a) == len(b)
    for (sa, ca), (sb, cb) in zip(a, b):
        assert sa == sb and ca["message"] == cb["message"] and len(ca["data"]) == len(cb["data"])
        for (x, px), (y, py) in zip(ca["data"], cb["data"]):
            assert px == py and abs(x - y) <= tol
    return True


def test_snapshot_round_trip_and_life_cycle(tmp_path):
    """create_snapshot keeps one file per collection (the old one is deleted first, app.py:130-141);
    recover_snapshot replaces the collection's points and leaves other collections alone."""
    rng = np.random.default_rng(3)
    dim = 16
    svc = _service(dim, 64)
    for c in ("S1_Employees", "S1_Customers"):
        svc.create_collection(c)
    vecs = rng.normal(size=(5, dim)).astype(np.float32)
    for i in range(4):
        assert svc.insert_point("S1_Employees", list(vecs[i]), f"id{i}", f"n{i}", "S1", time_created=f"t{i}")[0] == 201
    svc.insert_point("S1_Customers", list(vecs[4]), "c0", "cust", "S1", time_created="t9")
    st, info = svc.create_snapshot("S1_Employees", str(tmp_path))
    assert st == 200 and info["points"] == 4
    st, info2 = svc.create_snapshot("S1_Employees", str(tmp_path))
    snaps = svc.list_snapshots("S1_Employees", str(tmp_path))
    assert [x["name"] for x in snaps] == [info2["name"]]                 # the first one was deleted
    assert svc.create_snapshot("nope_Employees", str(tmp_path))[0] == 404
    before = [svc.search_point("S1_Employees", list(vecs[i]), "S1") for i in range(4)]
    # mutate, then recover
    svc.delete_point("S1_Employees", "id1")
    svc.insert_point("S1_Employees", list(vecs[4]), "idX", "x", "S1")
    assert svc.search_point("S1_Employees", list(vecs[1]), "S1") != before[1]
    path = str(tmp_path / "S1_Employees" / info2["name"])
    assert svc.recover_snapshot("S1_Employees", path) == (200, {"message": "Recover snapshot successfully"})
    after = [svc.search_point("S1_Employees", list(vecs[i]), "S1") for i in range(4)]
    assert _same_answers(after, before)
    assert svc.point_count("S1_Employees") == 4 and svc.point_count("S1_Customers") == 1
    assert svc.search_point("S1_Employees", list(vecs[4]), "S1")[1]["data"] == [] or \
        svc.search_point("S1_Employees", list(vecs[4]), "S1")[1]["data"][0][1]["id"] != "idX"
    # into a service that has never seen the collection (app.py:177-188 creates it)
    svc2 = _service(dim, 64)
    assert svc2.recover_snapshot("S1_Employees", path)[0] == 200
    assert _same_answers([svc2.search_point("S1_Employees", list(vecs[i]), "S1") for i in range(4)], before)
    assert svc2.recover_snapshot("Employees", path)[0] == 404            # name must be <store>_<role>
    assert svc2.recover_snapshot("S1_Employees", str(tmp_path / "missing.npz"))[0] == 404


def test_database_client_reshapes_like_the_reference():
    from frhot.database_client import DatabaseClient
    svc = _service(8, 32)
    cli = DatabaseClient(svc)

    async def run():
        assert await cli.ensure_collections_exist("S9") is True
        assert svc.collection_exists("S9_Customers") and svc.collection_exists("S9_Employees")
        v = [1.0] + [0.0] * 7
        assert await cli.insert_point("S9_Employees", v, "7", "Sev", "S9") is True
        assert await cli.insert_point("missing", v, "7", "Sev", "S9") is False
        res = await cli.search_point("S9_Employees", v, "S9")
        assert len(res) == 1 and res[0]["id"] == "7" and res[0]["payload"]["name"] == "Sev" and res[0]["score"] > 0.99
        assert await cli.search_point("missing", v, "S9") == []          # every failure -> []
        assert await cli.search_point("S9_Employees", [0.0] * 7 + [1.0], "S9") == []
        assert await cli.delete_point("S9_Employees", "7") is True
        assert await cli.search_point("S9_Employees", v, "S9") == []
        assert await cli.delete_point("missing", "7") is False
    asyncio.run(run())


# --------------------------------------------------------------- handlers ---
def _b64(img):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img[:, :, ::-1]).save(buf, format="PNG")
    return base64.b64encode(buf.getvalue()).decode()


@pytest.fixture(params=["seams", "fused"])
def client(request):
    """The two routes over CPU stand-ins, served through the two seams (get_embedding + search_point)
    and through the fused step (RecogniseBatcher -> GalleryService.recognise_points)."""
    from fastapi import FastAPI
    from fastapi.testclient import TestClient
    from fakes import FakeContext
    from frhot import legacy
    from frhot.api import build_router
    from frhot.batcher import RecogniseBatcher
    from frhot.database_client import DatabaseClient
    from frhot.face_service import FaceService
    legacy.install_engine(HashEmbedder())
    svc = _service(4096, 64)
    recogniser = None
    if request.param == "fused":
        svc.gallery.ctx = FakeContext()
        recogniser = RecogniseBatcher(svc, 16, max_delay_s=0.005)
    app = FastAPI()
    app.include_router(build_router(FaceService(DatabaseClient(svc), recogniser=recogniser, max_in_flight=16), api_key="k"))
    with TestClient(app) as c:
        c.svc = svc
        yield c
    if recogniser is not None:
        recogniser.close()
    legacy.install_engine(None)


def test_handlers_status_table(client):
    from frhot import synth
    rng = np.random.default_rng(1)
    ann, bob = _b64(synth.crop(rng, 90, 80)), _b64(synth.crop(rng, 70, 100))
    H = {"X-API-Key": "k"}
    r = client.post("/face_recog_img_base64", json={"img_base64": ann, "role": "1", "store_id": "T"})
    assert r.status_code == 403                                            # routes.py:21-28
    r = client.post("/face_recog_img_base64", json={"img_base64": ann, "role": "1", "store_id": "T"}, headers=H)
    assert r.status_code == 404 and r.json() == {
        "status": 0, "message": "Face is not existed! Please register your face or checkin again"}
    assert client.svc.collection_exists("T_Employees") and client.svc.collection_exists("T_Customers")
    # enrol: validation table (legacy.py:278-292)
    for body, msg in [({"img_base64": ann, "id": "", "name": "Ann", "role": "1", "store_id": "T"}, "id and name are required"),
                      ({"img_base64": "", "id": "1", "name": "Ann", "role": "1", "store_id": "T"}, "invalid"),
                      ({"img_base64": ann, "id": "1", "name": "Ann", "role": "5", "store_id": "T"}, "invalid"),
                      ({"img_base64": ann, "id": "1", "name": "Ann", "role": "1", "store_id": ""}, "store_id is required")]:
        r = client.post("/create_face_img_base64", json=body, headers=H)
        assert r.status_code == 400 and r.json() == {"status": 2, "message": msg}
    r = client.post("/create_face_img_base64", json={"img_base64": ann, "id": "345", "name": "Ann", "role": "1", "store_id": "T"}, headers=H)
    assert r.status_code == 201 and r.json() == {"status": 1, "message": "create face Ann with id 345 successfully"}
    r = client.post("/create_face_img_base64", json={"img_base64": ann, "id": "999", "name": "Dup", "role": "1", "store_id": "T"}, headers=H)
    assert r.status_code == 409 and r.json() == {"status": 0, "message": "Face is existed! Please use another face"}
    r = client.post("/create_face_img_base64", json={"img_base64": ann, "id": "345", "name": "Ann", "role": "1", "store_id": "T", "is_update": True}, headers=H)
    assert r.status_code == 201 and r.json()["message"] == "update face Ann with id 345 successfully"
    # recognise: the enrolled face, a stranger, another collection, another store
    r = client.post("/face_recog_img_base64", json={"img_base64": ann, "role": "1", "store_id": "T"}, headers=H)
    assert r.status_code == 200 and r.json() == {"status": 1, "id": "345", "name": "Ann"}
    r = client.post("/face_recog_img_base64", json={"img_base64": bob, "role": "1", "store_id": "T"}, headers=H)
    assert r.status_code == 404
    r = client.post("/face_recog_img_base64", json={"img_base64": ann, "role": "0", "store_id": "T"}, headers=H)
    assert r.status_code == 404                                            # Customers collection is empty
    r = client.post("/face_recog_img_base64", json={"img_base64": ann, "role": "1", "store_id": "U"}, headers=H)
    assert r.status_code == 404
    r = client.post("/face_recog_img_base64", json={"img_base64": "", "role": "1", "store_id": "T"}, headers=H)
    assert r.status_code == 400 and r.json() == {"status": 2, "message": "invalid"}
    r = client.post("/face_recog_img_base64", json={"img_base64": "not-an-image", "role": "1", "store_id": "T"}, headers=H)
    assert r.status_code == 500 and r.json()["status"] == 2


# ----------------------------------------------------------- quality gates ---
def _mesh(turn=1.0, ear=0.3):
    """468 landmarks with the nose tip `turn` times nearer to the left face edge than to the right
    and both eyes open by aspect ratio `ear`."""
    from frhot import quality
    lm = np.full((468, 3), 0.5)
    lm[234, :2], lm[454, :2] = (0.2, 0.5), (0.8, 0.5)
    lm[5, :2] = (0.2 + 0.6 / (1.0 + turn), 0.5)
    for eye, cx in ((quality.LEFT_EYE_LANDMARKS, 0.35), (quality.RIGHT_EYE_LANDMARKS, 0.65)):
        lm[eye[0], :2], lm[eye[3], :2] = (cx - 0.05, 0.4), (cx + 0.05, 0.4)
        for top, bottom, dx in ((eye[1], eye[5], -0.02), (eye[2], eye[4], 0.02)):
            lm[top, :2], lm[bottom, :2] = (cx + dx, 0.4 - 0.05 * ear), (cx + dx, 0.4 + 0.05 * ear)
    return lm


def test_quality_gates_equal_the_oracle():
    from frhot import quality
    from oracle import quality as oq
    rng = np.random.default_rng(5)
    try:
        for turn in (0.2, 0.39, 0.41, 1.0, 2.4, 2.6, 7.0):
            for ear in (0.05, 0.199, 0.2, 0.201, 0.4):
                lm = _mesh(turn, ear) + rng.normal(0, 1e-4, (468, 3))
                quality.install_models(face_mesh=lambda img: [lm])
                assert quality.check_face_left_right(None) == oq.check_face_left_right([lm])
                assert quality.check_eyes_open(None) == oq.check_eyes_open([lm])
                assert quality.DetectDirection(lm) == oq.detect_direction(lm)
                for eye in (quality.LEFT_EYE_LANDMARKS, quality.RIGHT_EYE_LANDMARKS):
                    assert quality.eye_aspect_ratio(eye, lm) == pytest.approx(oq.eye_aspect_ratio(list(eye), lm), rel=1e-12)
        assert quality.DetectDirection(_mesh(2.6)) == "right" and quality.DetectDirection(_mesh(1 / 2.6)) == "left"
        quality.install_models(face_mesh=lambda img: [])
        assert quality.check_face_left_right(None) == oq.check_face_left_right([]) == (False, "Face not detected! Please try again")
        assert quality.check_eyes_open(None) is False
        img = np.zeros((120, 80, 3), np.uint8)
        for _ in range(200):
            dets = [rng.uniform(0.0, 1.08, (6, 2)) for _ in range(rng.integers(0, 3))]
            quality.install_models(face_detection=lambda im: dets)
            assert quality.is_full_face(img) == oq.is_full_face(img.shape, dets)
        seen = set()
        for _ in range(200):
            faces = [{"confidence": float(rng.choice([0.5, 0.79, 0.8, 0.95])), "is_real": bool(rng.random() < 0.8),
                      "facial_area": {"x": int(rng.integers(0, 50)), "y": int(rng.integers(0, 50)),
                                      "w": int(rng.integers(20, 200)), "h": int(rng.integers(20, 200)),
                                      "left_eye": (1, 2), "right_eye": (3, 4)}} for _ in range(rng.integers(0, 4))]
            quality.install_models(extract_faces=lambda im: faces)
            got, want = quality.detect_face(img), oq.detect_face(faces, img.shape[1])
            assert got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
            seen.add(len(got[0]))
        assert seen >= {0, 1, 2}
        assert quality.distance_face_to_camera((10, 0, 110, 0), 640) == oq.distance_face_to_camera((10, 0, 110, 0), 640) \
            == pytest.approx(14.3 * (4 * 640 / 4.8) / 100)
        quality.install_models()
        with pytest.raises(RuntimeError):
            quality.check_eyes_open(img)
        with pytest.raises(RuntimeError):
            quality.check_detect_blur(img)
    finally:
        quality.install_models()


@pytest.mark.parametrize("mode", ["seams", "fused"])
def test_check_in_gates_status_table(mode):
    """face_service.py:73-219,238-245: every gate's refusal, in the reference's order, through both
    service paths; enrolment and customers (is_checkin False) pass ungated."""
    from fastapi import FastAPI
    from fastapi.testclient import TestClient
    from fakes import FakeContext
    from frhot import legacy, quality, synth
    from frhot.api import build_router
    from frhot.batcher import RecogniseBatcher
    from frhot.database_client import DatabaseClient
    from frhot.face_service import FaceService
    rng = np.random.default_rng(2)
    sharp = synth.crop(rng, 200, 200)
    sharp[::2, ::2] = 255 - sharp[::2, ::2]                        # plenty of Laplacian energy
    blurred = np.full((200, 200, 3), 90, np.uint8)
    blurred[:, :, 1] += np.arange(200, dtype=np.uint8)[None, :] // 8
    state = {"mesh": [_mesh()], "faces": None, "dets": [np.full((6, 2), 0.5)]}

    def extract(img):
        if state["faces"] == "raise":
            raise ValueError("Face could not be detected")
        if state["faces"] is not None:
            return state["faces"]
        h, w = img.shape[:2]      # a 30-cm-wide face box at 40 cm
        bw = int(14.3 * (4 * w / 4.8) / 40.0)
        return [{"confidence": 0.93, "is_real": True,
                 "facial_area": {"x": (w - bw) // 2, "y": 0, "w": bw, "h": h, "left_eye": None, "right_eye": None}}]

    ctx = FakeContext()
    legacy.install_engine(HashEmbedder())
    quality.install_engine(ctx)
    quality.install_models(face_mesh=lambda im: state["mesh"], face_detection=lambda im: state["dets"], extract_faces=extract)
    svc = _service(4096, 64)
    recogniser = None
    if mode == "fused":
        svc.gallery.ctx = ctx
        recogniser = RecogniseBatcher(svc, 16, max_delay_s=0.002)
    app = FastAPI()
    app.include_router(build_router(FaceService(DatabaseClient(svc), recogniser=recogniser, max_in_flight=16,
                                                quality_gates=True), api_key="k"))
    H = {"X-API-Key": "k"}
    try:
        with TestClient(app) as c:
            def recog(img, role="1"):
                r = c.post("/face_recog_img_base64", json={"img_base64": _b64(img), "role": role, "store_id": "Q"}, headers=H)
                return r.status_code, r.json()
            r = c.post("/create_face_img_base64", headers=H,
                       json={"img_base64": _b64(sharp), "id": "7", "name": "Ann", "role": "1", "store_id": "Q"})
            assert r.status_code == 201
            assert recog(sharp) == (200, {"status": 1, "id": "7", "name": "Ann"})
            table = [
                ({"mesh": [_mesh(turn=3.0)]}, sharp, "Face is looking right! Please look straight"),
                ({"mesh": [_mesh(turn=1 / 3.0)]}, sharp, "Face is looking left! Please look straight"),
                ({"mesh": []}, sharp, "Face not detected! Please try again"),
                ({"mesh": [_mesh(ear=0.1)]}, sharp, "Eyes are closed! Please open your eyes"),
                ({"faces": [{"confidence": 0.9, "is_real": True, "facial_area": {"x": 0, "y": 0, "w": 200, "h": 200}}]},
                 sharp, "Face is too close! Please move back"),
                ({"faces": [{"confidence": 0.9, "is_real": True, "facial_area": {"x": 90, "y": 90, "w": 25, "h": 25}}]},
                 sharp, "Face is too far! Please move forward"),
                ({"dets": [np.array([[0.5, 0.5], [0.5, 0.5], [0.5, 0.5], [0.5, 1.2], [0.5, 0.5], [0.5, 0.5]])]},
                 sharp, "Your mouth is not detected! Please show your face"),
                ({"dets": []}, sharp, "Face is not detected"),
                ({}, blurred, "Face is blur! Please keep your face in focus"),
            ]
            for change, img, message in table:
                saved = dict(state)
                state.update(change)
                assert recog(img) == (400, {"status": 2, "message": message}), message
                state.update(saved)
            # detector failure and no usable face at check-in: the reference's 500 bodies
            state["faces"] = "raise"
            assert recog(sharp) == (500, {"status": 2, "message": "Error when detecting face! Please try again"})
            state["faces"] = [{"confidence": 0.5, "is_real": True, "facial_area": {"x": 0, "y": 0, "w": 50, "h": 50}}]
            assert recog(sharp) == (500, {"status": 2, "message": "Error when processing face! Please try again"})   # argmin of an empty list (face_service.py:147,261-273)
            # the spoof verdict of the whole image (legacy.py:60-74): the most confident face decides
            state["faces"] = None
            real_extract = extract

            def spoof_second_pass(img, calls=[0]):
                calls[0] += 1
                out = real_extract(img)
                if calls[0] % 2 == 0:       # first call: detect_face; second: get_embedding's own pass
                    out = out + [{"confidence": 0.99, "is_real": False, "facial_area": out[0]["facial_area"]}]
                return out
            quality.install_models(face_mesh=lambda im: state["mesh"], face_detection=lambda im: state["dets"],
                                   extract_faces=spoof_second_pass)
            assert recog(sharp) == (400, {"status": 2, "message": "Face is not real! Please use your real face"})
            quality.install_models(face_mesh=lambda im: state["mesh"], face_detection=lambda im: state["dets"],
                                   extract_faces=extract)
            # customers are not gated (is_checkin False): the blurred image goes through to the search
            assert recog(blurred, role="0")[0] == 404
            # enrolment is not gated either; a detector failure there (and for customers) answers the same
            # 500 as at check-in: the reference's "no embedding" branch returns a name it has just
            # deleted (face_service.py:117,126) and lands in the handler of :133-143
            state.update(mesh=[_mesh(turn=5.0)])
            r = c.post("/create_face_img_base64", headers=H,
                       json={"img_base64": _b64(blurred), "id": "8", "name": "Bob", "role": "1", "store_id": "Q"})
            assert r.status_code == 201
            state["faces"] = "raise"
            r = c.post("/create_face_img_base64", headers=H,
                       json={"img_base64": _b64(sharp[:100]), "id": "9", "name": "Cy", "role": "1", "store_id": "Q"})
            assert r.status_code == 500 and r.json() == {"status": 2, "message": "Error when detecting face! Please try again"}
            assert recog(sharp, role="0") == (500, {"status": 2, "message": "Error when detecting face! Please try again"})
            # nothing was stored for id 9
            state["faces"] = None
            assert not any(p["payload"]["id"] == "9" for p in svc._points.values())
            # a detector that finds no face in the whole image during the spoof pass: 500, as the
            # reference's unguarded face_is_real[0] (legacy.py:74) inside the embedding try block
            def empty_second_pass(img, calls=[0]):
                calls[0] += 1
                return extract(img) if calls[0] % 2 == 1 else []
            quality.install_models(face_mesh=lambda im: state["mesh"], face_detection=lambda im: state["dets"],
                                   extract_faces=empty_second_pass)
            state.update(mesh=[_mesh()])
            assert recog(sharp) == (500, {"status": 2, "message": "Error! Please try again"})
    finally:
        if recogniser is not None:
            recogniser.close()
        legacy.install_engine(None)
        quality.install_engine(None)
        quality.install_models()


# ------------------------------------------------------- weights / synth ---
def test_weight_blob_layout(weights_f32):
    from frhot.weights import LAYERS, MAGIC, pack_blob, quantize_fp16
    blob = pack_blob(weights_f32)
    assert bytes(blob[:8]) == MAGIC
    version, count = struct.unpack("<II", bytes(blob[8:16]))
    assert version == 1 and count == 2 * len(LAYERS)
    entries = {}
    for i in range(count):
        name, dtype, ndim, d0, d1, d2, d3, off, nbytes = struct.unpack("<16sII4IQQ", bytes(blob[16 + 56 * i:16 + 56 * (i + 1)]))
        entries[name.rstrip(b"\0").decode()] = (dtype, ndim, (d0, d1, d2, d3), off, nbytes)
        assert off % 64 == 0 and off + nbytes <= blob.size
    dt, nd, dims, off, nb = entries["conv3_2.w"]
    assert (dt, nd, dims, nb) == (0, 4, (256, 3, 3, 256), 256 * 9 * 256 * 2)
    w = blob[off:off + nb].view(np.float16).reshape(256, 3, 3, 256)
    assert np.array_equal(w.astype(np.float32), quantize_fp16(weights_f32)["conv3_2.w"])
    dt, nd, dims, off, nb = entries["fc7.b"]
    assert (dt, nd, dims[0], nb) == (1, 1, 4096, 4096 * 4)
    assert entries["fc6.w"][2] == (4096, 7, 7, 512) and entries["fc7.w"][2][:2] == (4096, 4096)
    conv_w = sum(int(np.prod(weights_f32[n + ".w"].shape)) for n, *_ in LAYERS[:13])
    conv_b = sum(int(weights_f32[n + ".b"].size) for n, *_ in LAYERS[:13])
    assert conv_w + conv_b == 14_714_688
    assert weights_f32["fc6.w"].size + 4096 == 102_764_544 and weights_f32["fc7.w"].size + 4096 == 16_781_312


def test_synth_is_deterministic_and_packs():
    from frhot import synth
    a, b = synth.crops(5, 3, 40, 60), synth.crops(5, 3, 40, 60)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    buf, off, hw, st = synth.pack_crops(a)
    for im, o, (h, w), s in zip(a, off, hw, st):
        assert s == 3 * w and np.array_equal(buf[o:o + h * w * 3].reshape(h, w, 3), im) and o % 16 == 0
    g = synth.unit_gallery(1, 50, 64, sparse=True)
    assert np.allclose(np.linalg.norm(g, axis=1), 1, atol=1e-5) and (g >= 0).all()
    q = synth.planted_queries(np.random.default_rng(0), g[:10], 0.8)
    cos = (q * g[:10]).sum(1)
    assert 0.6 < cos.mean() < 0.95


def test_micro_batcher_groups_concurrent_callers():
    from frhot.batcher import EmbedBatcher

    class Ctx:
        max_batch = 8
        sizes = []

        def embed(self, images, gamma=False):
            self.sizes.append(len(images))
            return np.stack([np.full(4096, float(im[0, 0, 0]) + (100 if gamma else 0), np.float32) for im in images])

    ctx = Ctx()
    b = EmbedBatcher(ctx, max_delay_s=0.05)
    out = {}

    def call(i):
        out[i] = b.embed(np.full((4, 4, 3), i, np.uint8), gamma=(i % 2 == 1))
    ts = [threading.Thread(target=call, args=(i,)) for i in range(20)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    b.close()
    for i in range(20):
        assert out[i][0] == i + (100 if i % 2 else 0)
    assert sum(ctx.sizes) == 20 and max(ctx.sizes) <= 8 and len(ctx.sizes) < 20


def test_search_batcher_groups_concurrent_searches():
    """Searches in flight together share one scan; every caller gets the answer the unbatched
    service gives, including the validation failures and the empty-collection answer."""
    from frhot.batcher import SearchBatcher
    from frhot.database_client import DatabaseClient
    svc = _service(64, 256)
    rng = np.random.default_rng(4)
    vecs = rng.standard_normal((40, 64)).astype(np.float32)
    for col in ("A_Employees", "A_Customers", "B_Employees"):
        svc.create_collection(col)
    for i in range(40):
        col = ("A_Employees", "A_Customers")[i % 2]
        assert svc.insert_point(col, list(vecs[i]), str(i), f"n{i}", "A")[0] == 201
    calls = []
    real_search = svc.gallery.search
    svc.gallery.search = lambda q, qkeys=None, k=5: (calls.append(len(q)), real_search(q, qkeys, k))[1]
    reqs = [(("A_Employees", "A_Customers")[i % 2], list(vecs[i]), "A") for i in range(40)]
    reqs += [("B_Employees", list(vecs[0]), "B"), ("missing", list(vecs[0]), "A"), ("A_Employees", [0.0] * 3, "A")]
    want = [svc.search_point(*r) for r in reqs]
    assert want[0][1]["message"] == "Point found" and want[0][1]["data"][0][1]["id"] == "0"
    assert want[40] == (200, {"message": "Point not found", "data": []}) and want[41][0] == 404 and want[42][0] == 404
    n_unbatched = len(calls)
    calls.clear()
    sb = SearchBatcher(svc, max_batch=64, max_delay_s=0.05)
    cli = DatabaseClient(svc, sb)

    async def run():
        return await asyncio.gather(*[cli.search_point(*r) for r in reqs])
    got = asyncio.run(run())
    sb.close()
    ref = [DatabaseClient._reshape(*w) for w in want]
    assert len(got) == len(ref)
    for g, r in zip(got, ref):          # the stand-in's BLAS rounds a batch differently in the last bit
        assert [(x["id"], x["payload"]) for x in g] == [(x["id"], x["payload"]) for x in r]
        assert np.allclose([x["score"] for x in g], [x["score"] for x in r], atol=1e-6)
    assert sum(calls) == n_unbatched and len(calls) < 5      # 40 real scans' worth of queries in a few calls
    assert sb.items == len(reqs) and sb.batches < 5


def test_keras_h5_converter_layout(monkeypatch):
    """from_keras_h5 against a stand-in for h5py that serves a Keras-style file: kernels stored
    [kh, kw, Cin, Cout] come out [Cout, kh, kw, Cin], in layer order, fc8 ignored, fc7 as a matrix.
    (h5py and vgg_face_weights.h5 do not exist in this image; the stand-in has h5py's File / group /
    attrs shape and nothing else.)"""
    import sys
    import types
    from frhot.weights import LAYERS, from_keras_h5, pack_blob

    rng = np.random.default_rng(0)
    store, names = {}, []

    def add(name, arrays):
        names.append(name.encode())
        g = types.SimpleNamespace(attrs={"weight_names": [f"{name}/{k}".encode() for k in arrays]}, data={
            f"{name}/{k}": v for k, v in arrays.items()})
        store[name] = g

    class Group:
        def __init__(self, g):
            self.attrs, self._d = g.attrs, g.data

        def __getitem__(self, k):
            return self._d[k]

    class File:
        def __init__(self, path, mode):
            self.attrs = {"layer_names": names}

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def __getitem__(self, k):
            return Group(store[k])

    add("zero_padding2d_1", {})                                   # layers without weights are skipped
    kernels = {}
    for (name, kh, kw, cin, cout) in LAYERS:
        if name == "fc6":                                         # 411 MB as float32: a repeating pattern
            k = np.broadcast_to(rng.standard_normal((kh, kw, cin, 1), dtype=np.float32), (kh, kw, cin, cout))
        else:
            k = rng.standard_normal((kh, kw, cin, cout), dtype=np.float32)
        kernels[name] = k
        add(name, {"kernel:0": k, "bias:0": np.arange(cout, dtype=np.float32)})
        add(name + "_relu", {})
    add("fc8", {"kernel:0": np.zeros((1, 1, 4096, 2622), np.float32), "bias:0": np.zeros(2622, np.float32)})
    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=File))
    out = from_keras_h5("vgg_face_weights.h5")
    assert out["conv1_1.w"].shape == (64, 3, 3, 3) and out["fc6.w"].shape == (4096, 7, 7, 512)
    assert out["fc7.w"].shape == (4096, 4096)
    k = kernels["conv3_2"]
    assert out["conv3_2.w"][17, 2, 0, 100] == k[2, 0, 100, 17] and out["conv3_2.w"][255, 1, 1, 0] == k[1, 1, 0, 255]
    assert out["fc6.w"][9, 6, 5, 300] == kernels["fc6"][6, 5, 300, 9]
    assert out["fc7.w"][5, 77] == kernels["fc7"][0, 0, 77, 5]
    assert np.array_equal(out["conv5_3.b"], np.arange(512, dtype=np.float32))
    # a Keras convolution with kernel k equals the oracle's convolution with the converted weight
    import torch
    import torch.nn.functional as F
    x = torch.from_numpy(rng.standard_normal((1, 64, 6, 6), dtype=np.float32))
    kk = kernels["conv1_2"]
    keras = F.conv2d(x, torch.from_numpy(np.ascontiguousarray(kk.transpose(3, 2, 0, 1))), padding=1)    # HWIO -> OIHW
    ours = F.conv2d(x, torch.from_numpy(out["conv1_2.w"]).permute(0, 3, 1, 2), padding=1)               # KRSC -> OIHW
    assert torch.equal(keras, ours)
    assert pack_blob(out).size > 268_000_000


# ------------------------------------------- failure isolation (host side) ---
def test_one_bad_crop_fails_only_its_own_request():
    """A crop the library would refuse (wrong channel count, a strip that resizes to nothing) fails
    at submit(), alone; a refusal that only the device call reports makes the batch go out again
    crop by crop, so the other requests of the group still get their answers."""
    from frhot.batcher import EmbedBatcher, crop_problem

    class Ctx:
        max_batch = 8
        calls = []

        def embed(self, images, gamma=False):
            self.calls.append(len(images))
            if any(im[0, 0, 0] == 13 for im in images):
                raise RuntimeError("frhot error -1: bad crop")
            return np.stack([np.full(4096, float(im[0, 0, 0]), np.float32) for im in images])

    assert crop_problem(np.zeros((1, 300, 3), np.uint8)) is not None          # int(1 * 224/300) == 0
    assert crop_problem(np.zeros((300, 1, 3), np.uint8)) is not None
    assert crop_problem(np.zeros((5, 5, 4), np.uint8)) is not None
    assert crop_problem(np.zeros((5, 5), np.uint8)) is not None
    assert crop_problem(np.zeros((5, 5, 3), np.float32)) is not None
    assert crop_problem(np.zeros((1, 224, 3), np.uint8)) is None
    ctx = Ctx()
    b = EmbedBatcher(ctx, max_delay_s=0.2)
    crops = [np.full((4, 4, 3), i, np.uint8) for i in range(6)]
    crops[2] = np.zeros((1, 300, 3), np.uint8)         # refused at submit
    crops[4] = np.full((4, 4, 3), 13, np.uint8)        # refused by the device call only
    futs = [b.submit(c) for c in crops]
    res = []
    for f in futs:
        try:
            res.append(f.result(timeout=10)[0])
        except Exception as e:
            res.append(type(e).__name__)
    b.close()
    assert res == [0.0, 1.0, "ValueError", 3.0, "RuntimeError", 5.0]
    assert ctx.calls[0] == 5 and sorted(ctx.calls[1:]) == [1, 1, 1, 1, 1]


def _b64_pil(img, mode=None, fmt="PNG"):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img, mode=mode).save(buf, format=fmt)
    return base64.b64encode(buf.getvalue()).decode()


def test_decoder_shapes_and_channel_outcomes(client):
    """decode_image shapes its output as cv2.imdecode(flags=-1) does; what the reference cannot embed
    (4 channels, grey) ends in its 500 branch (face_service.py:246-253), also here."""
    from frhot.face_service import FaceService, decode_image
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 255, size=(40, 30, 3), dtype=np.uint8)
    rgba = np.dstack([rgb, np.full((40, 30), 200, np.uint8)])
    grey = rgb[:, :, 0]
    g16 = (grey.astype(np.uint16) * 257)
    d = decode_image(_b64_pil(rgb))
    assert d.shape == (40, 30, 3) and np.array_equal(d, rgb[:, :, ::-1])
    d = decode_image(_b64_pil(rgba))
    assert d.shape == (40, 30, 4) and np.array_equal(d[:, :, :3], rgb[:, :, ::-1]) and (d[:, :, 3] == 200).all()
    assert decode_image(_b64_pil(grey)).shape == (40, 30)
    d = decode_image(_b64_pil(g16))
    assert d.dtype == np.uint16 and np.array_equal(d, g16)
    H = {"X-API-Key": "k"}
    for img in (_b64_pil(rgba), _b64_pil(grey), _b64_pil(g16)):
        r = client.post("/face_recog_img_base64", json={"img_base64": img, "role": "1", "store_id": "T"}, headers=H)
        assert r.status_code == 500 and r.json() == {"status": 2, "message": "Error! Please try again"}
    # an injected decoder replaces PIL
    seen = []
    svc = FaceService(None, decoder=lambda b: seen.append(b) or rgb)
    assert svc.decoder("xyz") is rgb and seen == ["xyz"]


def test_failed_writes_leak_no_rows_and_keep_the_live_collection(tmp_path):
    """insert_point hands its row back when the device write fails; recover_snapshot writes the
    snapshot before it drops anything, so a failing write leaves the collection answering."""
    svc = _service(8, 16)
    svc.create_collection("S_Employees")
    rng = np.random.default_rng(5)
    vecs = rng.normal(size=(6, 8)).astype(np.float32)
    for i in range(6):
        assert svc.insert_point("S_Employees", vecs[i], f"p{i}", f"n{i}", "S")[0] == 201
    st, info = svc.create_snapshot("S_Employees", str(tmp_path))
    path = str(tmp_path / "S_Employees" / info["name"])
    before = [svc.search_point("S_Employees", vecs[i], "S") for i in range(6)]
    real_upsert, real_write = svc.gallery.upsert, svc.gallery.write_f16

    def boom(*a, **k):
        raise RuntimeError("device write failed")

    svc.gallery.upsert = boom
    free0 = svc._rows.free_count()
    assert svc.insert_point("S_Employees", vecs[0], "q", "q", "S") == (404, {"message": "device write failed"})
    assert svc._rows.free_count() == free0
    svc.gallery.upsert = real_upsert
    svc.gallery.write_f16 = boom
    assert svc.recover_snapshot("S_Employees", path)[0] == 404
    assert svc.point_count("S_Employees") == 6
    assert [svc.search_point("S_Employees", vecs[i], "S") for i in range(6)] == before
    assert svc._rows.free_count() == 16 - 6
    svc.gallery.write_f16 = real_write
    assert svc.recover_snapshot("S_Employees", path)[0] == 200
    assert svc.point_count("S_Employees") == 6 and svc._rows.free_count() == 10
    # a snapshot larger than the gallery is refused before anything is dropped
    small = _service(8, 4)
    assert small.recover_snapshot("S_Employees", path)[0] == 404 and small.point_count("S_Employees") == 0


def test_recognise_batcher_groups_and_isolates_failures():
    """Requests in flight together share one fused call; a crop the library refuses fails alone."""
    from fakes import FakeContext
    from frhot.batcher import RecogniseBatcher
    svc = _service(4096, 64)
    ctx = FakeContext(max_batch=8)
    svc.gallery.ctx = ctx
    svc.create_collection("S_Employees")
    crops = [np.full((6, 6, 3), i, np.uint8) for i in range(5)]
    for i in range(3):
        emb = ctx.embedder.embed(crops[i], gamma=True)
        assert svc.insert_point("S_Employees", emb, f"p{i}", f"n{i}", "S")[0] == 201
    real = ctx.recognise

    def picky(gallery, images, *a, **k):
        if any(im[0, 0, 0] == 4 for im in images):
            raise RuntimeError("frhot error -1: bad crop")
        return real(gallery, images, *a, **k)

    ctx.recognise = picky
    b = RecogniseBatcher(svc, 8, max_delay_s=0.2)
    futs = [b.submit("S_Employees", crops[i], "S", want_embedding=(i == 1)) for i in range(5)]
    futs.append(b.submit("S_Employees", np.zeros((1, 400, 3), np.uint8), "S"))
    futs.append(b.submit("nope", crops[0], "S"))
    res = []
    for f in futs:
        try:
            res.append(f.result(timeout=10))
        except Exception as e:
            res.append(type(e).__name__)
    b.close()
    for i in range(3):
        st, content, emb = res[i]
        assert st == 200 and content["message"] == "Point found" and content["data"][0][1]["id"] == f"p{i}"
        assert (emb is not None) == (i == 1)
    assert res[3][:2] == (200, {"message": "Point not found", "data": []})
    assert res[4] == "RuntimeError" and res[5] == "ValueError"
    assert res[6][0] == 404
    assert ctx.calls == [1] * 5          # the refused group call, then one call per request (the refused one records nothing)


def test_row_allocator_keeps_a_keys_rows_together():
    """Rows of one key come from tiles of its own while unclaimed tiles last (a by-key search then
    reads few tiles); after that rows are borrowed, released rows go back to their tile's owner, and
    the count of free rows is exact throughout."""
    from frhot.gallery_service import TILE, _RowAllocator
    al = _RowAllocator(3 * TILE + 10)
    assert al.free_count() == 3 * TILE + 10
    a = [al.alloc(7) for _ in range(5)]
    b = [al.alloc(9) for _ in range(3)]
    assert a == [0, 1, 2, 3, 4] and b == [TILE, TILE + 1, TILE + 2]
    a += [al.alloc(7) for _ in range(TILE)]                      # spills into a second tile of its own
    assert {r // TILE for r in a} == {0, 2}
    c = [al.alloc(11) for _ in range(10)]                        # the short last tile
    assert c == list(range(3 * TILE, 3 * TILE + 10))
    assert al.free_count() == 3 * TILE + 10 - len(a) - len(b) - len(c)
    d = al.alloc(11)                                             # every tile is claimed: a borrowed row
    assert d // TILE in (1, 2) and d not in a + b
    al.release(a[0])
    assert al.alloc(7) == a[0]                                   # a released row goes back to its owner
    al.release(d)
    free = al.free_count()
    got = [al.alloc(13) for _ in range(free)]                    # a key without tiles drains everything left
    assert len(set(got)) == free and not set(got) & set(a + b + c)
    with pytest.raises(MemoryError):
        al.alloc(13)


def test_crop_args_one_row_view_gets_a_usable_stride():
    """A one-row image keeps whatever row stride its view had (numpy ignores the stride of a
    length-one axis, also in ascontiguousarray): the marshalled stride is the row length."""
    from frhot.engine import _crop_args
    one = np.ascontiguousarray(np.zeros((1, 7, 3), np.uint8)[::-1])
    assert one.strides[0] < 0
    rev = np.zeros((5, 4, 3), np.uint8)[::-1]
    wide = np.zeros((6, 10, 3), np.uint8)[:, 2:7]
    _keep, _ptrs, hw, strides = _crop_args([one, rev, wide])
    assert hw.tolist() == [[1, 7], [5, 4], [6, 5]] and strides.tolist() == [21, 12, 30]


def test_search_does_not_hold_the_table_lock_while_the_device_works():
    """An enrolment and a deletion complete while a search's scan is in flight (the reference's DB
    service serves its handlers concurrently, qdrant_database_FE/app.py:242-438); a row that was
    freed and written again during the scan is not reported with its new owner's payload."""
    import threading
    dim = 64
    svc = _service(dim, 256)
    gal = svc.gallery
    col = "7_Employees"
    assert svc.create_collection(col)[0] == 201
    rng = np.random.default_rng(4)
    va, vb, vc = (rng.standard_normal(dim).astype(np.float32) for _ in range(3))
    assert svc.insert_point(col, va, "A", "Ann", 7)[0] == 201
    assert svc.insert_point(col, vb, "B", "Bob", 7)[0] == 201
    started, go = threading.Event(), threading.Event()
    plain_search = gal.search

    def slow_search(q, qkeys=None, k=5):
        out = plain_search(q, qkeys, k)         # the scan sees Ann's vector in her row ...
        started.set()
        assert go.wait(10)                      # ... and is still "on the device" while the table changes
        return out

    gal.search = slow_search
    answer = {}
    t = threading.Thread(target=lambda: answer.update(res=svc.search_point(col, va, 7)))
    t.start()
    assert started.wait(10)
    # while the scan is in flight: Ann is deleted and Cid enrolled, into the row Ann had
    assert svc.delete_point(col, "A") == (200, {"message": "Point deleted"})
    assert svc.insert_point(col, vc, "C", "Cid", 7)[0] == 201
    assert svc.point_count(col) == 2
    go.set()
    t.join(10)
    assert not t.is_alive()
    status, content = answer["res"]
    assert status == 200
    # Ann's row scored 1.0 for the scan, but it belongs to Cid now: nobody is reported for it
    assert all(p["id"] not in ("A", "C") for _score, p in content["data"])
    gal.search = plain_search
    status, content = svc.search_point(col, vc, 7)
    assert status == 200 and content["data"] and content["data"][0][1]["id"] == "C"
    # two searches at once: neither holds the lock while its scan runs (both reach the gallery before either returns)
    inside, both = [], threading.Event()

    def meet(q, qkeys=None, k=5):
        inside.append(1)
        if len(inside) == 2:
            both.set()
        assert both.wait(10), "the second search never reached the gallery while the first was in it"
        return plain_search(q, qkeys, k)

    gal.search = meet
    ts = [threading.Thread(target=svc.search_point, args=(col, vb, 7)) for _ in range(2)]
    for th in ts:
        th.start()
    for th in ts:
        th.join(15)
    assert not any(th.is_alive() for th in ts) and both.is_set()
