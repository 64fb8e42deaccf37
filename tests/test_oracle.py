"""CPU tests of the oracle itself: known answers derived from the reference's own lines
(gamma LUT formula, vote rule, eviction policy) and internal consistency of the
third-party restatements.  No GPU."""
import json
import os

import numpy as np
import pytest

from oracle import preprocess as op
from oracle import search as osr
from oracle import vggface

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_gamma_table_known_values():
    """legacy.py:108-109 with gamma=1.5: table[i] = uint8((i/255)^(2/3) * 255), truncated."""
    t = op.gamma_table(1.5)
    assert t.dtype == np.uint8 and t.shape == (256,)
    assert t[0] == 0 and t[255] == 255
    assert t[64] == int((64 / 255.0) ** (1 / 1.5) * 255) == 101
    assert t[128] == 161
    assert (np.diff(t.astype(int)) >= 0).all()
    gold = np.load(os.path.join(GOLDEN, "gamma_lut_1p5.npy"))
    assert np.array_equal(t, gold)


def test_crop_box_is_python_slice():
    img = np.arange(10 * 12 * 3, dtype=np.uint8).reshape(10, 12, 3)
    c = op.crop_box(img, (2.7, 1.2, 5.1, 6.9))   # ints: x1=2,y1=1,x2=int(7.8)=7,y2=int(8.1)=8
    assert c.shape == (7, 5, 3)
    assert np.array_equal(c, img[1:8, 2:7])


def test_fit_size_truncates_like_deepface():
    assert op.fit_size(224, 224) == (224, 224)
    assert op.fit_size(448, 448) == (224, 224)
    assert op.fit_size(130, 517) == (224, 56)          # int(130 * 224/517) = int(56.32)
    assert op.fit_size(601, 97) == (36, 224)
    dw, dh = op.fit_size(3, 7)
    assert dw == 224 and dh == 96
    # truncation can lose a pixel where the real product is 224 - epsilon
    for h in range(100, 700, 7):
        for w in range(100, 700, 11):
            dw, dh = op.fit_size(h, w)
            assert 0 < dw <= 224 and 0 < dh <= 224 and max(dw, dh) >= 223


def test_resize_identity_constant_and_area():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(224, 224, 3), dtype=np.uint8)
    assert np.array_equal(op.resize_linear_u8(img, 224, 224), img)          # scale 1: taps hit pixel centres
    const = np.full((57, 91, 3), 137, np.uint8)
    assert (op.resize_linear_u8(const, 200, 123) == 137).all()              # weights sum to 2048
    big = rng.integers(0, 256, size=(40, 60, 3), dtype=np.uint8)
    a = op.resize_linear_u8(big, 30, 20)                                    # exact 2x -> box average
    s = big.astype(int)
    want = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(a, want.astype(np.uint8))


def test_resize_matches_float_bilinear_within_one_lsb():
    """The fixed-point path is a rounding of true bilinear interpolation at half-pixel centres."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(83, 131, 3), dtype=np.uint8)
    dw, dh = 224, 141
    got = op.resize_linear_u8(img, dw, dh).astype(np.float64)
    sx = (np.arange(dw) + 0.5) * (131 / dw) - 0.5
    sy = (np.arange(dh) + 0.5) * (83 / dh) - 0.5
    x0 = np.floor(sx).astype(int)
    y0 = np.floor(sy).astype(int)
    fx = sx - x0
    fy = sy - y0
    fx = np.where((x0 < 0) | (x0 >= 130), 0.0, fx)
    x0c, x1c = np.clip(x0, 0, 130), np.clip(x0 + 1, 0, 130)
    y0c, y1c = np.clip(y0, 0, 82), np.clip(y0 + 1, 0, 82)
    f = img.astype(np.float64)
    top = f[y0c][:, x0c] * (1 - fx)[None, :, None] + f[y0c][:, x1c] * fx[None, :, None]
    bot = f[y1c][:, x0c] * (1 - fx)[None, :, None] + f[y1c][:, x1c] * fx[None, :, None]
    ref = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    assert np.abs(got - ref).max() <= 1.0


def test_resize_agrees_with_an_independent_bilinear():
    """torch's bilinear interpolation (half-pixel centres, edge clamp, no antialiasing: the convention
    cv2.INTER_LINEAR follows) is code this repository did not write; the fixed-point restatement stays
    within one grey level of it for enlargements and for reductions other than the exact 2x area case."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(7)
    for (h, w, dh, dw) in [(83, 131, 141, 224), (300, 180, 224, 134), (224, 97, 224, 97), (500, 640, 175, 224),
                           (31, 29, 224, 209), (449, 447, 224, 223)]:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        got = op.resize_linear_u8(img, dw, dh).astype(np.float64)
        t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
        ref = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        assert np.abs(got - ref).max() <= 1.0, (h, w, dh, dw, np.abs(got - ref).max())


def test_edge_geometries_agree_with_an_independent_bilinear():
    """The crop geometries the GPU parity test walks (tests/test_gpu_hotpath.py::_crops: square, wide,
    tall, one pixel, one row, almost-2x, almost-identity, more than 6x down, ...) through deepface's
    aspect fit, against torch's bilinear interpolation: within one grey level everywhere.  The exact
    2x reduction (448 x 448) is the documented exception: cv2 averages 2 x 2 boxes there, and the
    box average equals bilinear taps at weight 1/2 each, so it too stays within one level."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    shapes = [(96, 300), (300, 96), (171, 204), (257, 113), (299, 298), (448, 448), (224, 224), (130, 517), (601, 97),
              (64, 80), (50, 50), (1, 1), (2, 3), (3, 500), (447, 449), (225, 223), (1500, 1400), (300, 200)]
    assert len(shapes) == 18
    for h, w in shapes:
        img = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        dw, dh = op.fit_size(h, w)
        got = op.resize_linear_u8(img, dw, dh).astype(np.float64)
        t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
        ref = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False, antialias=False)[0].permute(1, 2, 0).numpy()
        assert got.shape == ref.shape and np.abs(got - ref).max() <= 1.0, (h, w, dh, dw, np.abs(got - ref).max())


def test_pad_is_centred_and_becomes_minus_mean():
    img = np.full((100, 50, 3), 200, np.uint8)
    u8 = op.resize_pad_u8(img)
    assert u8.shape == (224, 224, 3)
    dw, dh = op.fit_size(100, 50)
    left = (224 - dw) // 2
    assert (u8[:, :left] == 0).all() and (u8[:, left + dw:] == 0).all() and (u8[:, left:left + dw] == 200).all()
    x = op.preprocess(img, flip=True)
    assert x.dtype == np.float32
    assert np.allclose(x[0, 0], -np.asarray(op.VGGFACE2_MEAN, np.float32))
    # float round trip of the reference: u8/255*255 in float32, then minus the mean
    want = np.float32(np.float32(np.float32(200) / np.float32(255)) * np.float32(255)) - np.float32(op.VGGFACE2_MEAN[1])
    assert x[0, left, 1] == want


def test_flip_and_mean_channel_order():
    img = np.zeros((224, 224, 3), np.uint8)
    img[..., 0] = 10   # B
    img[..., 2] = 30   # R
    x = op.preprocess(img, flip=True)
    # after the reverse, channel 0 holds R and gets mean[0]
    assert np.isclose(x[5, 5, 0], 30 - 91.4953, atol=1e-4) and np.isclose(x[5, 5, 2], 10 - 131.0912, atol=1e-4)
    y = op.preprocess(img, flip=False)
    assert np.isclose(y[5, 5, 0], 10 - 91.4953, atol=1e-4)


def test_dark_image_skips_the_division():
    """deepface resize_image divides by 255 only when img.max() > 1; normalize_input multiplies regardless."""
    img = np.ones((64, 64, 3), np.uint8)
    x = op.preprocess(img)
    assert np.isclose(x[100, 100, 0], 255.0 - 91.4953, atol=1e-3)
    img2 = np.full((64, 64, 3), 2, np.uint8)
    x2 = op.preprocess(img2)
    assert np.isclose(x2[100, 100, 0], 2.0 - 91.4953, atol=1e-3)


def test_vggface_shapes_and_homogeneity(weights_q):
    """post-ReLU fc7, 4096-d, non-negative; scaling input and biases by s scales the output by s."""
    rng = np.random.default_rng(2)
    x = rng.normal(0, 60, size=(1, 224, 224, 3)).astype(np.float32)
    y = vggface.forward(x, weights_q)
    assert y.shape == (1, 4096) and (y >= 0).all() and y.max() > 0
    s = np.float32(1 / 16)
    ws = {k: (v * s if k.endswith(".b") else v) for k, v in weights_q.items()}
    ys = vggface.forward(x * s, ws)
    assert np.allclose(ys / s, y, rtol=2e-3, atol=2e-3 * y.max())
    e = vggface.l2_normalize(y)
    assert np.allclose(np.linalg.norm(e, axis=1), 1.0)
    assert np.array_equal(vggface.l2_normalize(np.zeros(8)), np.zeros(8))   # zero vector unchanged


def test_macs_per_face():
    macs = 0
    hw = 224
    for _name, cin, cout, pool in vggface.CONV_LAYERS:
        macs += hw * hw * 9 * cin * cout
        if pool:
            hw //= 2
    macs += 7 * 7 * 512 * 4096 + 4096 * 4096
    assert macs == vggface.MACS_PER_FACE == 15_466_168_320


def test_cosine_topk_against_brute_force():
    rng = np.random.default_rng(3)
    g = rng.normal(size=(700, 64)).astype(np.float32)
    q = rng.normal(size=(9, 64)).astype(np.float32)
    keys = rng.integers(0, 3, size=700).astype(np.int32)
    keys[5] = -1
    qk = np.array([0, 1, 2, -2, 0, 1, 2, 7, -2], np.int32)
    s, r = osr.cosine_topk(q, g, 5, keys=keys, qkeys=qk, chunk=128)
    gn = g / np.linalg.norm(g, axis=1, keepdims=True)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    full = qn @ gn.T
    for i in range(9):
        ok = (keys >= 0) if qk[i] == -2 else (keys == qk[i])
        cand = np.where(ok)[0]
        order = cand[np.lexsort((cand, -full[i, cand]))][:5]
        want = np.full(5, -1)
        want[:len(order)] = order
        assert np.array_equal(r[i], want)
    assert (r[7] == -1).all() and np.isneginf(s[7]).all()
    s2, r2 = osr.cosine_topk_fast(q, osr.unit_rows(g), 5)
    s3, r3 = osr.cosine_topk(q, g, 5)
    assert np.array_equal(r2, r3) and np.allclose(s2, s3, atol=1e-6)


def _p(i):
    return {"id": i, "name": f"n{i}", "store_id": "S", "time_created": "t"}


def test_vote_truth_table():
    """Cases read off qdrant_database_FE/app.py:340-384 (threshold 0.54)."""
    with open(os.path.join(GOLDEN, "vote_cases.json")) as f:
        cases = json.load(f)
    assert len(cases) >= 8
    for c in cases:
        hits = [(s, _p(i)) for s, i in c["hits"]]
        out = osr.vote(hits)
        if c["want"] is None:
            assert out == {"message": "Point not found", "data": []}, c
        else:
            assert out["message"] == "Point found", c
            (score, payload), = out["data"]
            assert payload["id"] == c["want"][1] and abs(score - c["want"][0]) < 1e-12, c


def test_collections_eviction_and_delete():
    """app.py:207-240: the 10th insert of one id deletes the oldest so that 9 + 1 remain; delete removes all."""
    col = osr.Collections(dim=8)
    assert col.create_collection("S_Employees") == (201, {"message": "Collection created"})
    assert col.create_collection("S_Employees")[0] == 200
    rng = np.random.default_rng(4)
    for i in range(12):
        st, _ = col.insert_point("S_Employees", rng.normal(size=8), "A", "Ann", "S", f"2024/01/01 00:00:{i:02d}")
        assert st == 201
    pts = [p for p in col.cols["S_Employees"] if p[2]["id"] == "A"]
    assert len(pts) == 10
    assert sorted(p[2]["time_created"][-2:] for p in pts) == [f"{i:02d}" for i in range(2, 12)]
    assert col.insert_point("nope", np.zeros(8), "A", "Ann", "S", "t")[0] == 404
    assert col.insert_point("S_Employees", np.zeros(7), "A", "Ann", "S", "t")[0] == 404
    assert col.insert_point("S_Employees", np.zeros(8), "", "Ann", "S", "t")[0] == 404
    assert col.delete_point("S_Employees", "A") == (200, {"message": "Point deleted"})
    assert not col.cols["S_Employees"]
    assert col.delete_point("S_Employees", "")[0] == 404


def test_collections_search_filters_store_and_votes():
    col = osr.Collections(dim=4)
    col.create_collection("S_Customers")
    v = np.array([1, 0, 0, 0], np.float32)
    col.insert_point("S_Customers", v, "A", "Ann", "S", "t1")
    col.insert_point("S_Customers", v, "B", "Bob", "OTHER", "t2")      # same collection, other store
    st, out = col.search_point("S_Customers", v, "S")
    assert st == 200 and out["message"] == "Point found" and out["data"][0][1]["id"] == "A"
    st, out = col.search_point("S_Customers", np.array([0, 1, 0, 0], np.float32), "S")
    assert out == {"message": "Point not found", "data": []}
    assert col.search_point("S_Customers", np.zeros(3), "S")[0] == 404
    assert col.search_point("missing", v, "S")[0] == 404


def test_oracle_fixtures_unchanged():
    """The committed preprocessing / search vectors (outputs of this oracle) still reproduce."""
    z = np.load(os.path.join(GOLDEN, "preprocess_oracle.npz"))
    for i in range(3):
        assert np.array_equal(op.preprocess_u8(z[f"crop{i}"], apply_gamma=True, flip=True), z[f"u8_{i}"])
    z = np.load(os.path.join(GOLDEN, "search_oracle.npz"))
    s, r = osr.cosine_topk(z["q"], z["g"].astype(np.float32), 5, keys=z["keys"], qkeys=z["qkeys"],
                           gallery_dtype=np.float16)
    assert np.array_equal(r, z["rows"]) or np.abs(s - z["scores"]).max() < 1e-3


def test_topk_stored_equals_cosine_topk():
    """The large-gallery form of the search oracle (stored fp16 rows scored as they stand, only the
    rows at or above a chunk's k-th best kept) gives the answer of the plain form: keys, ANY,
    tombstones, a key nobody has, exact ties across chunks, fewer rows than k."""
    from frhot import synth
    rng = np.random.default_rng(0)
    n, d = 150_000, 64
    g = synth.unit_gallery(1, n, d)
    g[140_000] = g[5]
    g[70_000] = g[5]
    keys = rng.integers(0, 3, n).astype(np.int32)
    keys[::50] = -1
    t = rng.integers(0, n, 40)
    q = synth.planted_queries(rng, g[t], 0.7)
    q[0] = g[5]
    qk = keys[t].copy()
    qk[::4] = -2
    qk[3] = 77
    q16 = osr.unit_rows(q).astype(np.float16)
    g16 = osr.unit_rows(g).astype(np.float16)
    for k, kw in ((10, dict(keys=keys, qkeys=qk)), (5, {})):
        a_s, a_r = osr.cosine_topk(q, g, k, gallery_dtype=np.float16, **kw)
        b_s, b_r = osr.topk_stored(q16, g16, k, **kw)
        assert np.array_equal(a_r, b_r)
        assert np.array_equal(a_s, b_s)
    assert list(b_r[0, :3]) == [5, 70_000, 140_000]
    s, r = osr.topk_stored(q16[:2], g16[:3], 5, row0=100)
    assert (r[:, 3:] == -1).all() and np.isneginf(s[:, 3:]).all() and sorted(r[0, :3]) == [100, 101, 102]


def test_blur_gate_hand_cases_and_fixture():
    """The Laplacian stencil and border rule on cases worked by hand, 1-pixel-wide images, the
    grey conversion's rounding, and the committed oracle outputs."""
    import json
    from oracle import quality as oq
    with open(os.path.join(GOLDEN, "blur_hand_cases.json")) as f:
        for case in json.load(f):
            grey = np.array(case["grey"], np.uint8)
            img = np.repeat(grey[:, :, None], 3, axis=2)      # B = G = R = v  ->  grey v (weights sum to 2^15)
            assert np.array_equal(oq.bgr2gray(img), grey)
            assert oq.laplacian_sums(img) == (case["sum"], case["sumsq"])
            n = grey.size
            assert oq.laplacian_var(img) == pytest.approx(case["sumsq"] / n - (case["sum"] / n) ** 2, rel=1e-12)
    # grey weights: 0.114 / 0.587 / 0.299 in 15 bits, round half up
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 1, 1], [10, 200, 77]]], np.uint8)
    assert oq.bgr2gray(px).tolist() == [[29, 150, 76, 1, (10 * 3735 + 200 * 19235 + 77 * 9798 + 16384) >> 15]]
    assert np.array_equal(oq.bgr2gray(np.dstack([px, np.full((1, 5, 1), 9, np.uint8)])), oq.bgr2gray(px))   # alpha ignored
    assert oq.laplacian_sums(np.full((1, 1, 3), 200, np.uint8)) == (0, 0)
    assert oq.check_detect_blur(np.full((40, 40, 3), 128, np.uint8)) is False
    z = np.load(os.path.join(GOLDEN, "blur_oracle.npz"))
    for i in range(5):
        assert oq.laplacian_sums(z[f"crop{i}"]) == tuple(int(v) for v in z[f"sums{i}"])
        assert oq.laplacian_var(z[f"crop{i}"]) == float(z[f"var{i}"])
