"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle."""
import numpy as np
import pytest

from conftest import cosine_rows

pytestmark = pytest.mark.gpu


def _crops():
    from frhot import synth
    imgs = synth.crops(seed=11, n=5, lo=96, hi=300)
    rng = np.random.default_rng(5)
    imgs.append(synth.crop(rng, 448, 448))      # exact 2x downscale -> area path
    imgs.append(synth.crop(rng, 224, 224))      # identity resize
    imgs.append(synth.crop(rng, 130, 517))      # wide: vertical padding
    imgs.append(synth.crop(rng, 601, 97))       # tall: horizontal padding, truncated dsize
    imgs.append(np.ones((64, 80, 3), np.uint8)) # max <= 1: DeepFace skips the /255
    imgs.append(np.zeros((50, 50, 3), np.uint8))
    imgs.append(synth.crop(rng, 1, 1))          # one pixel
    imgs.append(synth.crop(rng, 2, 3))
    imgs.append(synth.crop(rng, 3, 500))        # fits to one row of 224 pixels, 223 rows of padding
    imgs.append(synth.crop(rng, 447, 449))      # almost the 2x case: must take the bilinear path
    imgs.append(synth.crop(rng, 225, 223))      # almost the identity
    imgs.append(synth.crop(rng, 1500, 1400))    # more than 6x down: plain bilinear taps, no area averaging
    imgs.append(np.full((300, 200, 3), 255, np.uint8))
    return imgs


@pytest.mark.parametrize("gamma,flip", [(False, True), (True, True), (True, False)])
def test_preprocess_exact(gpu_ctx, gamma, flip):
    from oracle import preprocess as op
    crops = _crops()
    for lo in range(0, len(crops), 9):
        _check_preprocess(gpu_ctx, op, crops[lo:lo + 9], gamma, flip)


def _check_preprocess(gpu_ctx, op, imgs, gamma, flip):
    gpu_ctx.embed(imgs, gamma=gamma, flip=flip)
    u8 = gpu_ctx.debug_read(0, len(imgs))
    x0 = gpu_ctx.debug_read(1, len(imgs))
    for i, im in enumerate(imgs):
        want_u8 = op.preprocess_u8(im, apply_gamma=gamma, flip=flip)
        assert np.array_equal(u8[i], want_u8), f"crop {i} {im.shape}: resized bytes differ"
        want = (op.preprocess(im, apply_gamma=gamma, flip=flip) * np.float32(1.0 / 16.0)).astype(np.float16)
        assert np.array_equal(x0[i, 1:225, 1:225, :3], want), f"crop {i}: network input differs"
        assert not x0[i, 0].any() and not x0[i, 225].any() and not x0[i, :, 0].any() and not x0[i, :, 225].any()
        assert not x0[i, :, :, 3].any()


def _oracle_layers(x_nhwc, weights):
    """Per-layer float32 activations (NHWC, after pooling where present) + fc6, fc7."""
    import torch
    import torch.nn.functional as F
    from oracle.vggface import CONV_LAYERS
    outs = []
    x = torch.from_numpy(x_nhwc).permute(0, 3, 1, 2)
    with torch.no_grad():
        for name, _ci, _co, pool in CONV_LAYERS:
            w = torch.from_numpy(weights[name + ".w"]).permute(0, 3, 1, 2)
            x = F.relu(F.conv2d(x, w, torch.from_numpy(weights[name + ".b"]), padding=1))
            if pool:
                x = F.max_pool2d(x, 2, 2)
            outs.append(x.permute(0, 2, 3, 1).numpy())
        w6 = torch.from_numpy(weights["fc6.w"]).permute(0, 3, 1, 2)
        x = F.relu(F.conv2d(x, w6, torch.from_numpy(weights["fc6.b"])))
        outs.append(x.flatten(1).numpy())
    return outs


def test_layers_match_oracle(gpu_ctx, weights_q):
    """Every conv layer and fc6, against the oracle run on the same fp16-rounded weights."""
    from frhot import synth
    from oracle import preprocess as op
    imgs = synth.crops(seed=3, n=2, lo=150, hi=260)
    gpu_ctx.embed(imgs)
    x = np.stack([op.preprocess(im) for im in imgs])
    want = _oracle_layers(x, weights_q)
    s = 1.0 / 16.0
    for l in range(13):
        got = gpu_ctx.debug_read(10 + l, len(imgs)).astype(np.float32) / s
        if l < 12:
            assert not got[:, 0].any() and not got[:, -1].any() and not got[:, :, 0].any() and not got[:, :, -1].any(), \
                f"layer {l}: halo not zero"
            got = got[:, 1:-1, 1:-1]
        ref = want[l]
        assert got.shape == ref.shape
        err = np.abs(got - ref).max()
        scale = np.abs(ref).max()
        assert err <= 4e-3 * scale, f"conv layer {l}: max err {err} vs scale {scale}"
    got6 = gpu_ctx.debug_read(30, len(imgs)).astype(np.float32) / s
    assert np.abs(got6 - want[13]).max() <= 4e-3 * np.abs(want[13]).max()


def test_first_layer_output_is_not_kept_in_device_memory(weights_f32):
    """The embed path computes conv1_1 inside conv1_2's kernel: a context holds no buffer for its
    output (6.4 MB per face of max_batch) until frhot_debug_read is asked for stage 10, which then
    makes it with the stand-alone kernel."""
    import torch
    from frhot import synth
    from frhot.engine import Context
    from frhot.weights import pack_blob
    n = 32
    per_face = 226 * 226 * 64 * 2
    ctx = Context(device=0, max_batch=n)
    try:
        ctx.load_model(pack_blob(weights_f32), input_scale=1.0 / 16.0)
        imgs = synth.crops(seed=12, n=n, lo=120, hi=200)
        ctx.embed(imgs)
        ctx.embed(imgs)
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        ctx.embed(imgs)
        layer2 = ctx.debug_read(11, n)
        assert torch.cuda.mem_get_info()[0] == free0          # neither the embed nor a later stage allocates
        layer1 = ctx.debug_read(10, n)
        used = free0 - torch.cuda.mem_get_info()[0]
        assert abs(used - n * per_face) <= (32 << 20), used      # (the runtime hands out memory in pieces of its own size)
        assert layer1.shape == (n, 226, 226, 64) and layer1[:, 1:-1, 1:-1].any() and layer2.any()
        assert np.array_equal(ctx.debug_read(10, n), layer1)
    finally:
        ctx.close()


def test_embedding_cosine(gpu_ctx, weights_f32, weights_q):
    """cosine(GPU descriptor, oracle descriptor) >= 0.9999 per face (tolerance of the north star)."""
    from frhot import synth
    from oracle import preprocess as op, vggface
    imgs = synth.crops(seed=21, n=6, lo=112, hi=448)
    got = gpu_ctx.embed(imgs, gamma=True)
    assert got.shape == (6, 4096) and got.dtype == np.float32
    assert np.allclose(np.linalg.norm(got.astype(np.float64), axis=1), 1.0, atol=1e-5)
    assert (got >= 0).all()
    x = np.stack([op.preprocess(im, apply_gamma=True) for im in imgs])
    ref32 = vggface.embed(x, weights_f32)      # the reference's precision: float32 weights
    refq = vggface.embed(x, weights_q)         # same fp16-rounded weights as the GPU
    c32 = cosine_rows(got, ref32)
    cq = cosine_rows(got, refq)
    assert cq.min() >= 0.9999, f"cosine vs same-weights oracle {cq}"
    assert c32.min() >= 0.9999, f"cosine vs float32-weights oracle {c32}"


def test_embed_batch_invariance(gpu_ctx):
    """A crop's descriptor does not depend on its batch neighbours or position."""
    from frhot import synth
    imgs = synth.crops(seed=8, n=5, lo=100, hi=200)
    a = gpu_ctx.embed(imgs)
    b = gpu_ctx.embed(imgs[::-1])[::-1]
    c = gpu_ctx.embed([imgs[2]])
    assert np.array_equal(a, b)
    assert np.array_equal(a[2:3], c)


def test_embed_strided_crop_views(gpu_ctx):
    """The reference hands over `img[y1:y2, x1:x2]` (face_service.py:180-181): a view whose rows are
    a full image row apart.  Such views give the descriptors of their contiguous copies, also when
    the batch is large enough for the host staging to be packed by several threads."""
    from frhot import synth
    rng = np.random.default_rng(31)
    big = [synth.crop(rng, 700, 900) for _ in range(4)]
    views = []
    for i in range(24):
        h, w = int(rng.integers(400, 690)), int(rng.integers(500, 890))
        y, x = int(rng.integers(0, 700 - h + 1)), int(rng.integers(0, 900 - w + 1))
        views.append(big[i % 4][y:y + h, x:x + w])
    assert not views[0].flags["C_CONTIGUOUS"]
    assert sum(v.nbytes for v in views[:16]) > (12 << 20)      # first chunk: three or more packing threads
    a = gpu_ctx.embed(views, gamma=True)
    b = gpu_ctx.embed([np.ascontiguousarray(v) for v in views], gamma=True)
    assert np.array_equal(a, b)
    # the resized bytes of the last chunk (crops 16..23 at a batch limit of 16) against the oracle
    from oracle import preprocess as op
    assert gpu_ctx.max_batch == 16
    gpu_ctx.embed(views, gamma=True)
    u8 = gpu_ctx.debug_read(0, 8)
    for i in (0, 3, 7):
        want = op.preprocess_u8(np.ascontiguousarray(views[16 + i]), apply_gamma=True, flip=True)
        assert np.array_equal(u8[i], want), f"view {16 + i}"


def test_embed_full_batch(gpu_ctx, weights_f32):
    """The benchmark's batch (256 crops, several tiles per block in every layer, both tile walk
    directions, full-width channel tiles in the later layers): every descriptor equals, bit for bit,
    the one the same crop gets in a small batch (which runs those layers with the narrower channel
    tiles of small launches), and a sample meets the cosine tolerance against the oracle."""
    import torch
    from frhot import synth
    from frhot.engine import Context
    from frhot.weights import pack_blob
    from oracle import preprocess as op, vggface
    imgs = synth.crops(seed=77, n=256, lo=112, hi=300)
    big = Context(device=0, max_batch=256)
    big.load_model(pack_blob(weights_f32), input_scale=1.0 / 16.0)
    got = big.embed(imgs, gamma=True)
    # the device-pointer call runs its two halves side by side (the early layers with half of the
    # persistent blocks each) where the host call starts the second half when its crops arrive: same bits
    buf, offsets, hw, strides = synth.pack_crops(imgs)
    crops_dev = torch.from_numpy(buf).to("cuda:0")
    out = torch.empty((256, 4096), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    big.embed_dev(crops_dev.data_ptr(), offsets, hw, strides, Context.flags(gamma=True, flip=True), out.data_ptr(), None)
    big.sync()
    assert np.array_equal(out.cpu().numpy(), got)
    big.close()
    assert got.shape == (256, 4096) and np.isfinite(got).all()
    assert np.allclose(np.linalg.norm(got.astype(np.float64), axis=1), 1.0, atol=1e-5)
    pick = [0, 1, 15, 16, 100, 127, 128, 200, 254, 255]
    small = gpu_ctx.embed([imgs[i] for i in pick], gamma=True)
    assert np.array_equal(small, got[pick])
    sample = [0, 128, 255]
    x = np.stack([op.preprocess(imgs[i], apply_gamma=True) for i in sample])
    ref = vggface.embed(x, weights_f32)
    assert cosine_rows(got[sample], ref).min() >= 0.9999


def test_embed_largest_batch(weights_f32):
    """The largest batch a context accepts (2048 crops: layer inputs of up to 13 GB, the ring
    kernels' 32-bit input offsets at 3.5 of their 4 GiB): every descriptor equals, bit for bit,
    the one its crop gets in a batch of 32; one crop more is refused."""
    from frhot import synth
    from frhot._lib import FrhotError
    from frhot.engine import Context
    from frhot.weights import pack_blob
    base = synth.crops(seed=5, n=32, lo=112, hi=180)
    with pytest.raises(FrhotError):
        Context(device=0, max_batch=2049)
    big = Context(device=0, max_batch=2048)
    try:
        big.load_model(pack_blob(weights_f32), input_scale=1.0 / 16.0)
        want = big.embed(base, gamma=True)
        order = np.random.default_rng(9).permutation(2048) % 32
        got = big.embed([base[i] for i in order], gamma=True)
    finally:
        big.close()
    assert got.shape == (2048, 4096) and np.isfinite(got).all()
    assert np.array_equal(got, want[order])


def _check_topk(got_s, got_r, ref_s, ref_r, full_scores=None, tol=1e-4):
    """Identical id lists except inside score ties within tol."""
    assert got_r.shape == ref_r.shape
    for i in range(ref_r.shape[0]):
        if np.array_equal(got_r[i], ref_r[i]):
            continue
        for j in range(ref_r.shape[1]):
            if got_r[i, j] == ref_r[i, j]:
                continue
            assert (got_r[i, j] >= 0) == (ref_r[i, j] >= 0), f"query {i} slot {j}: presence differs"
            # a differing slot must hold a row whose oracle score ties the oracle's slot score
            assert full_scores is not None
            sc = full_scores(i, got_r[i, j]) if callable(full_scores) else full_scores[i, got_r[i, j]]
            assert abs(sc - ref_s[i, j]) <= tol, \
                f"query {i} slot {j}: row {got_r[i, j]} vs {ref_r[i, j]} is not a tie"
    ok = np.isfinite(ref_s)
    assert np.abs(got_s[ok] - ref_s[ok]).max() <= tol


@pytest.mark.parametrize("n,d,bq,k", [(5000, 512, 64, 10), (3001, 4096, 7, 5), (777, 512, 130, 5), (40, 512, 3, 5),
                                      (3, 64, 2, 5)])
def test_search_matches_oracle(gpu_ctx, n, d, bq, k):
    from frhot import synth
    from frhot.engine import Gallery
    from oracle import search as osr
    g = synth.unit_gallery(seed=n, n=n, d=d, sparse=(d == 4096))
    g[n // 2] = g[n // 3]                                    # exact duplicate -> documented tie
    rng = np.random.default_rng(n + 1)
    targets = rng.integers(0, n, size=bq)
    q = synth.planted_queries(rng, g[targets], 0.8)
    gal = Gallery(gpu_ctx, d, n + 10)
    gal.upsert(g, 0, np.arange(n))
    got_s, got_r = gal.search(q, None, k)
    ref_s, ref_r = osr.cosine_topk(q, g, k, gallery_dtype=np.float16)
    full = osr.unit_rows(q).astype(np.float16).astype(np.float32) @ osr.unit_rows(g).astype(np.float16).astype(np.float32).T
    _check_topk(got_s, got_r, ref_s, ref_r, full, tol=2e-5)
    # against the reference's float32 precision: ties within 1e-4
    ref_s32, ref_r32 = osr.cosine_topk(q, g, k)
    full32 = osr.unit_rows(q) @ osr.unit_rows(g).T
    _check_topk(got_s, got_r, ref_s32, ref_r32, full32, tol=1e-4)
    if n >= k:
        assert (got_r[:, 0] == targets).mean() >= 0.9 or d < 256
    gal.close()


@pytest.mark.parametrize("n,d,bq,k,with_keys", [(600_000, 64, 256, 5, False), (400_000, 128, 300, 10, True),
                                                 (300_000, 64, 64, 16, False), (500_000, 64, 256, 5, True),
                                                 (262_144 + 77, 192, 129, 5, False),
                                                 (2_200_000, 64, 50, 10, True), (280_000, 64, 2000, 5, False)])
def test_search_long_scan_matches_oracle(gpu_ctx, n, d, bq, k, with_keys):
    """Many gallery tiles per block (the running top-k lists, their pruning thresholds and the
    merge of 256 block lists all matter), every query-tile width, duplicates that tie, a key filter
    with tombstones, and a last tile that is partly past the end of the gallery.  Several
    cases give a block enough tiles (8 for the wide query tiles, 24 for the 64-query tile) to turn on the
    score bounds that the blocks share through device memory, among them one query tile with 256
    walkers and eight query tiles with 32."""
    from frhot import synth
    from frhot.engine import Gallery
    from oracle import search as osr
    rng = np.random.default_rng(n + d)
    g = rng.standard_normal((n, d), dtype=np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    dup = rng.integers(0, n, size=(200, 2))
    g[dup[:, 0]] = g[dup[:, 1]]                              # exact duplicates anywhere in the scan order
    targets = rng.integers(0, n, size=bq)
    q = synth.planted_queries(rng, g[targets], 0.7)
    q[-1] = g[dup[0, 1]]                                     # a query whose best two rows tie exactly
    keys = rng.integers(0, 3, size=n).astype(np.int32) if with_keys else None
    qkeys = None
    gal = Gallery(gpu_ctx, d, n)
    step = 100_000
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        gal.upsert(g[lo:hi], keys[lo:hi] if with_keys else 0, np.arange(lo, hi))
    if with_keys:
        qkeys = keys[targets].copy()
        qkeys[::7] = -2                                      # ANY
        dead = rng.integers(0, n, size=5000)
        gal.tombstone(dead)
        keys = keys.copy()
        keys[dead] = -1
    got_s, got_r = gal.search(q, qkeys, k)
    ref_s, ref_r = osr.cosine_topk(q, g, k, keys=keys, qkeys=qkeys, gallery_dtype=np.float16)
    qn = osr.unit_rows(q).astype(np.float16).astype(np.float32)
    # the stored rows are the oracle's fp16 unit rows, bit for bit
    sample = rng.integers(0, n, size=2000)
    stored, _ = gal.read(sample)
    assert np.array_equal(stored.view(np.uint16), osr.unit_rows(g[sample]).astype(np.float16).view(np.uint16))

    def score(i, row):
        return float(qn[i] @ osr.unit_rows(g[row:row + 1]).astype(np.float16).astype(np.float32)[0])

    _check_topk(got_s, got_r, ref_s, ref_r, score, tol=2e-5)
    assert (got_r >= 0).all()
    gal.close()


@pytest.mark.parametrize("k,d", [(10, 64), (5, 128), (16, 64)])
def test_search_many_equal_scores_in_one_tile(gpu_ctx, k, d):
    """More survivors of one query in one 256-row tile than the 64-query scan's selection queue holds
    (32): whole tiles of identical rows, once in a tile that is some block's FIRST (where the start
    lists are ranked, not inserted) and once in a later tile of a block, plus a run that straddles two
    tiles.  Equal scores come back in ascending row order, so the answer is exact and unique; queries
    that match nothing special ride along in the same query tile."""
    from frhot.engine import Gallery
    from oracle import search as osr
    rng = np.random.default_rng(k * 1000 + d)
    n = 150_000                                    # 586 tiles on 256 blocks: blocks walk 2 or 3 tiles
    g = rng.standard_normal((n, d), dtype=np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    va, vb, vc = (rng.standard_normal(d).astype(np.float32) for _ in range(3))
    g[5 * 256:6 * 256] = va                         # tile 5: block 5's first tile
    g[300 * 256:301 * 256] = vb                     # tile 300: block 44's second tile
    g[520 * 256 - 40:520 * 256 + 45] = vc           # 85 rows over the end of tile 519 and the start of tile 520
    bq = 64
    q = rng.standard_normal((bq, d), dtype=np.float32)
    q[3], q[17], q[40] = va, vb, vc
    q[41] = va + vb                                 # both tiles of equal rows score high for one query
    gal = Gallery(gpu_ctx, d, n)
    for lo in range(0, n, 50_000):
        gal.upsert(g[lo:lo + 50_000], 0, np.arange(lo, min(n, lo + 50_000)))
    got_s, got_r = gal.search(q, None, k)
    ref_s, ref_r = osr.cosine_topk(q, g, k, gallery_dtype=np.float16)
    assert np.array_equal(got_r[3], 5 * 256 + np.arange(k))
    assert np.array_equal(got_r[17], 300 * 256 + np.arange(k))
    assert np.array_equal(got_r[40], 520 * 256 - 40 + np.arange(k))
    assert np.array_equal(got_r, ref_r)
    assert np.abs(got_s - ref_s).max() <= 2e-5
    gal.close()


def test_search_key_filter_and_tombstones(gpu_ctx):
    from frhot import synth
    from frhot.engine import Gallery
    from oracle import search as osr
    n, d, bq, k = 4000, 512, 50, 5
    g = synth.unit_gallery(seed=77, n=n, d=d)
    rng = np.random.default_rng(78)
    keys = rng.integers(0, 6, size=n).astype(np.int32)
    targets = rng.integers(0, n, size=bq)
    q = synth.planted_queries(rng, g[targets], 0.8)
    qkeys = keys[targets].copy()
    qkeys[:5] = (qkeys[:5] + 1) % 6          # wrong collection: the planted row must not be found
    qkeys[5:8] = -2                          # ANY
    qkeys[8] = 99                            # a key nobody has -> empty answer
    gal = Gallery(gpu_ctx, d, n)
    gal.upsert(g, keys, np.arange(n))
    dead = targets[10:14]
    gal.tombstone(dead)
    keys_ref = keys.copy()
    keys_ref[dead] = -1
    got_s, got_r = gal.search(q, qkeys, k)
    ref_s, ref_r = osr.cosine_topk(q, g, k, keys=keys_ref, qkeys=qkeys, gallery_dtype=np.float16)
    full = osr.unit_rows(q).astype(np.float16).astype(np.float32) @ osr.unit_rows(g).astype(np.float16).astype(np.float32).T
    _check_topk(got_s, got_r, ref_s, ref_r, full, tol=2e-5)
    assert (got_r[8] == -1).all() and np.isneginf(got_s[8]).all()
    assert not np.isin(got_r[10:14, 0], dead).any()
    for i in range(5):
        assert targets[i] not in got_r[i]
    gal.close()


def test_search_by_key_equals_whole_scan(gpu_ctx):
    """A search whose queries carry keys may scan only the tiles that can hold each key (the reference
    searches one collection's own storage, qdrant_database_FE/app.py:316-337).  Whatever the way,
    the answer is the same, bit for bit: keys clustered in tiles of their own, keys scattered over
    every tile, a key spread one row per tile, rows that died or moved to another key, a key nobody
    has, ANY queries; also against the oracle, and after a fill whose keys were derived on the
    device (the tile lists are then rebuilt from device memory)."""
    from frhot import synth
    from frhot.engine import Gallery
    from oracle import search as osr
    n, d, k = 20_000, 128, 5
    rng = np.random.default_rng(2026)
    g = synth.unit_gallery(seed=91, n=n, d=d)
    keys = np.empty(n, np.int32)
    keys[:] = 100 + rng.integers(0, 5, size=n)                   # scattered: five keys over every tile
    for j in range(12):                                          # clustered: key j in rows 768 j .. of its own
        lo = 768 * j
        keys[lo:lo + 500 + 20 * j] = j
    keys[np.arange(30, n, 256)] = 50                             # one row in every tile
    gal = Gallery(gpu_ctx, d, n)
    gal.upsert(g, keys, np.arange(n))
    dead = rng.choice(n, size=300, replace=False)
    gal.tombstone(dead)
    keys_ref = keys.copy()
    keys_ref[dead] = -1
    moved = np.setdiff1d(np.arange(0, 400), dead)[:50]           # rows of key 0 re-written under key 7
    gal.upsert(g[moved], np.full(moved.size, 7, np.int32), moved)
    keys_ref[moved] = 7
    bq = 300
    targets = rng.integers(0, n, size=bq)
    q = synth.planted_queries(rng, g[targets], 0.8)
    qkeys = np.where(keys_ref[targets] >= 0, keys_ref[targets], 3).astype(np.int32)
    qkeys[:20] = -2
    qkeys[20:25] = 999                                           # nobody has it
    qkeys[25:40] = 50
    qkeys[40:60] = 7
    out = {}
    for mode in (Gallery.SEARCH_WHOLE, Gallery.SEARCH_BY_KEY, Gallery.SEARCH_AUTO):
        gal.search_mode(mode)
        out[mode] = gal.search(q, qkeys, k)
    for mode in (Gallery.SEARCH_BY_KEY, Gallery.SEARCH_AUTO):
        assert np.array_equal(out[mode][1], out[Gallery.SEARCH_WHOLE][1])
        assert np.array_equal(out[mode][0], out[Gallery.SEARCH_WHOLE][0])
    got_s, got_r = out[Gallery.SEARCH_BY_KEY]
    ref_s, ref_r = osr.cosine_topk(q, g, k, keys=keys_ref, qkeys=qkeys, gallery_dtype=np.float16)
    full = osr.unit_rows(q).astype(np.float16).astype(np.float32) @ osr.unit_rows(g).astype(np.float16).astype(np.float32).T
    _check_topk(got_s, got_r, ref_s, ref_r, full, tol=2e-5)
    assert (got_r[20:25] == -1).all() and np.isneginf(got_s[20:25]).all()
    # one query, one key: the serving shape
    gal.search_mode(Gallery.SEARCH_BY_KEY)
    s1, r1 = gal.search(q[100:101], qkeys[100:101], k)
    assert np.array_equal(r1[0], got_r[100]) and np.array_equal(s1[0], got_s[100])
    gal.close()
    # keys derived on the device
    gal = Gallery(gpu_ctx, d, n)
    gal.fill_synthetic(seed=5, row0=0, n=n, nkeys=40, dup_every=0)
    rows16, rkeys = gal.read(targets)
    qk = rkeys.astype(np.int32)
    gal.search_mode(Gallery.SEARCH_WHOLE)
    sw, rw = gal.search(rows16.astype(np.float32), qk, k)
    gal.search_mode(Gallery.SEARCH_BY_KEY)
    sb, rb = gal.search(rows16.astype(np.float32), qk, k)
    assert np.array_equal(rw, rb) and np.array_equal(sw, sb)
    assert (sb[:, 0] >= 0.999).all()
    gal.close()
    # keys strewn over every tile (100 keys x 1172 tiles): the lists are given up, the answers stay right
    n2 = 300_000
    gal = Gallery(gpu_ctx, 64, n2)
    gal.fill_synthetic(seed=6, row0=0, n=n2, nkeys=100, dup_every=0)
    t2 = rng.integers(0, n2, size=40)
    rows16, rkeys = gal.read(t2)
    gal.search_mode(Gallery.SEARCH_BY_KEY)
    sb, rb = gal.search(rows16.astype(np.float32), rkeys.astype(np.int32), k)
    gal.search_mode(Gallery.SEARCH_WHOLE)
    sw, rw = gal.search(rows16.astype(np.float32), rkeys.astype(np.int32), k)
    assert np.array_equal(rw, rb) and np.array_equal(sw, sb) and (sb[:, 0] >= 0.999).all()
    gal.close()


@pytest.mark.parametrize("k,bq", [(1, 64), (5, 64), (10, 64), (5, 200), (16, 256)])
def test_search_by_key_long_walks_with_many_keys_in_one_query_tile(gpu_ctx, k, bq):
    """Keys strewn over every tile of a gallery long enough that a walker takes 24 tiles and more: the
    by-key launch then shares score bounds between blocks.  Most keys have one or two of the 64 queries,
    so every block's 64-query tile also holds queries of OTHER keys, which it scores without their key
    filter: those columns must not reach the other queries' bounds (a query then lost its answer).
    With 200 or 256 queries the last keys' tiles start near the end of the query block, and their
    columns past it read bounds too (the table has to reach 63 queries past the block).
    Same answers as the whole-gallery scan, and as the oracle."""
    from frhot.engine import Gallery
    from oracle import search as osr
    rng = np.random.default_rng(35)
    n, d = 109_897, 192
    g = rng.standard_normal((n, d), dtype=np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    keys = rng.integers(0, 50, size=n).astype(np.int32)
    targets = rng.integers(0, n, size=bq)
    q = g[targets] + 0.5 * rng.standard_normal((bq, d), dtype=np.float32) / np.sqrt(d)
    qkeys = keys[targets].copy()
    qkeys[::5] = -2
    gal = Gallery(gpu_ctx, d, n)
    gal.upsert(g, keys, np.arange(n))
    dead = rng.integers(0, n, size=n // 100)
    gal.tombstone(dead)
    keys[dead] = -1
    gal.search_mode(Gallery.SEARCH_WHOLE)
    s_w, r_w = gal.search(q, qkeys, k)
    gal.search_mode(Gallery.SEARCH_BY_KEY)
    s_k, r_k = gal.search(q, qkeys, k)
    assert np.array_equal(r_k, r_w) and np.array_equal(s_k, s_w)
    ref_s, ref_r = osr.cosine_topk(q, g, k, keys=keys, qkeys=qkeys, gallery_dtype=np.float16)
    assert (r_k[:, 0] >= 0).all()
    assert np.array_equal(r_k >= 0, ref_r >= 0)
    assert np.abs(np.where(r_k >= 0, s_k - ref_s, 0.0)).max() <= 2e-5
    gal.close()


def test_recognise_by_key_equals_whole_scan(gpu_ctx):
    """The fused call (crops in, answers out) with keys: by key and over the whole gallery give the
    same rows and scores, from two threads at once as well (each call's index arrays travel in its
    own staging slot)."""
    import threading
    from frhot import synth
    from frhot.engine import Gallery
    imgs = synth.crops(seed=41, n=12, lo=120, hi=220)
    emb = gpu_ctx.embed(imgs)
    n = 6000
    g = synth.unit_gallery(seed=42, n=n, d=4096, sparse=True)
    keys = (np.arange(n) // 700).astype(np.int32)                # keys 0..8 in runs of 700 rows
    for i in range(12):
        g[500 * i + 7] = emb[i]
    gal = Gallery(gpu_ctx, 4096, n)
    gal.upsert(g, keys, np.arange(n))
    qkeys = keys[[500 * i + 7 for i in range(12)]].copy()
    qkeys[3] = -2
    qkeys[5] = (qkeys[5] + 1) % 9                                # the wrong collection
    gal.search_mode(Gallery.SEARCH_WHOLE)
    sw, rw, _ = gpu_ctx.recognise(gal, imgs, qkeys, 5)
    gal.search_mode(Gallery.SEARCH_BY_KEY)
    sb, rb, _ = gpu_ctx.recognise(gal, imgs, qkeys, 5)
    assert np.array_equal(rw, rb) and np.array_equal(sw, sb)
    assert all(rb[i, 0] == 500 * i + 7 for i in range(12) if i != 5) and 500 * 5 + 7 not in rb[5]
    res = [None, None]

    def call(j):
        res[j] = gpu_ctx.recognise(gal, imgs[::-1] if j else imgs, qkeys[::-1].copy() if j else qkeys, 5)
    th = [threading.Thread(target=call, args=(j,)) for j in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert np.array_equal(res[0][1], rb) and np.array_equal(res[1][1], rb[::-1])
    gal.close()


def test_search_dev_keyed_one_shard(gpu_ctx):
    """frhot_search_dev_keyed (queries and answers on the device, keys on the host: what a row-sharded
    search with keys calls on every shard): the same answer by key and over the whole shard, the
    shard's row base added, two calls in a row without waiting in between."""
    import torch
    from frhot import FrhotError, synth
    from frhot.engine import Gallery
    n, d, k, base = 9000, 64, 5, 1_000_000
    g = synth.unit_gallery(seed=14, n=n, d=d)
    keys = (np.arange(n) // 900).astype(np.int32)
    gal = Gallery(gpu_ctx, d, n)
    gal.upsert(g, keys, np.arange(n))
    gal.tombstone(np.arange(5, n, 11))
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(15)
    t = rng.integers(0, n, size=70)
    t = t[t % 11 != 5]
    bq = len(t)
    q16 = torch.from_numpy(g[t].astype(np.float16)).to(dev)
    qk = keys[t].copy()
    qk[0] = -2
    qk[1] = 4000
    outs = []
    for mode in (Gallery.SEARCH_WHOLE, Gallery.SEARCH_BY_KEY):
        gal.search_mode(mode)
        s1 = torch.empty((bq, k), dtype=torch.float32, device=dev)
        r1 = torch.empty((bq, k), dtype=torch.int64, device=dev)
        s2 = torch.empty((8, k), dtype=torch.float32, device=dev)
        r2 = torch.empty((8, k), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        gal.search_dev_keyed(q16.data_ptr(), qk, bq, k, s1.data_ptr(), r1.data_ptr(), base)
        gal.search_dev_keyed(q16.data_ptr(), qk[:8], 8, k, s2.data_ptr(), r2.data_ptr(), base)
        gpu_ctx.sync()
        outs.append((s1.cpu().numpy(), r1.cpu().numpy(), s2.cpu().numpy(), r2.cpu().numpy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    s1, r1, s2, r2 = outs[1]
    assert np.array_equal(r2, r1[:8]) and np.array_equal(s2, s1[:8])
    assert (r1[2:, 0] == t[2:] + base).all() and r1[0, 0] == t[0] + base and (r1[1] == -1).all()
    live = r1[2:] >= 0
    assert (keys[(r1[2:] - base)[live]] == np.repeat(qk[2:], k)[live.ravel()]).all()
    sd = torch.empty((2, k), dtype=torch.float32, device=dev)
    rd = torch.empty((2, k), dtype=torch.int64, device=dev)
    with pytest.raises(FrhotError):
        gal.search_dev_keyed(q16.data_ptr(), np.array([0, -1], np.int32), 2, k, sd.data_ptr(), rd.data_ptr(), 0)
    gal.close()


def test_dead_key_as_query_key_matches_nothing(gpu_ctx):
    """A query key of -1 (the tombstone value) or any other negative key but ANY is no row key:
    the host forms refuse it, the device-pointer form answers with no row at all, so deleted
    faces can never come back through it."""
    import torch
    from frhot import FrhotError, synth
    from frhot.engine import Gallery
    n, d = 3000, 64
    g = synth.unit_gallery(seed=12, n=n, d=d)
    gal = Gallery(gpu_ctx, d, n)
    gal.upsert(g, (np.arange(n) % 4).astype(np.int32), np.arange(n))
    gal.tombstone(np.arange(0, n, 3))
    with pytest.raises(FrhotError):
        gal.search(g[:4], np.array([0, -1, 1, 2], np.int32), 5)
    with pytest.raises(FrhotError):
        gal.search(g[:4], np.array([0, 1, -7, 2], np.int32), 5)
    dev = torch.device("cuda", 0)
    q16 = torch.from_numpy(g[[0, 1, 1, 3, 4, 0]].astype(np.float16)).to(dev)
    qk = torch.tensor([-1, -2, 1, -5, -2147483648, 0], dtype=torch.int32, device=dev)
    s = torch.empty((6, 5), dtype=torch.float32, device=dev)
    r = torch.empty((6, 5), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    gal.search_dev(q16.data_ptr(), qk.data_ptr(), 6, 5, s.data_ptr(), r.data_ptr(), 0)
    gpu_ctx.sync()
    r, s = r.cpu().numpy(), s.cpu().numpy()
    for i in (0, 3, 4):
        assert (r[i] == -1).all() and np.isneginf(s[i]).all()
    assert r[1, 0] == 1 and r[2, 0] == 1                 # row 1 is live with key 1; ANY finds it too
    assert (r[5] % 3 != 0).all() and (r[5] % 4 == 0).all()    # row 0 is dead: live rows of key 0 only
    gal.close()


def test_synthetic_fill_bit_exact(gpu_ctx):
    from frhot.engine import Gallery
    from oracle import synth_gallery as sg
    d, n = 512, 3000
    gal = Gallery(gpu_ctx, d, n)
    gal.fill_synthetic(seed=0xFACE0002, row0=0, n=n, nkeys=7, dup_every=100)
    rows = np.concatenate([np.arange(0, 300), np.arange(n - 200, n)])
    got, got_keys = gal.read(rows)
    want = sg.rows_f16(0xFACE0002, rows, d, dup_every=100)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    assert np.array_equal(got_keys, sg.keys(0xFACE0002, rows, 7, dup_every=100))
    assert np.array_equal(got[99].view(np.uint16), got[98].view(np.uint16))   # planted duplicate
    gal.close()


def test_recognise_fused_equals_two_calls(gpu_ctx):
    from frhot import synth
    from frhot.engine import Gallery
    imgs = synth.crops(seed=31, n=4, lo=120, hi=240)
    emb = gpu_ctx.embed(imgs)
    rng = np.random.default_rng(1)
    g = synth.unit_gallery(seed=32, n=600, d=4096, sparse=True)
    g[17] = emb[1]
    g[333] = emb[3]
    gal = Gallery(gpu_ctx, 4096, 600)
    gal.upsert(g, 0, np.arange(600))
    s1, r1 = gal.search(emb, None, 5)
    s2, r2, e2 = gpu_ctx.recognise(gal, imgs, None, 5, want_embeddings=True)
    assert np.array_equal(e2, emb)
    assert np.array_equal(r1, r2)
    assert np.abs(s1 - s2).max() <= 2e-3      # fp16 query rounding differs (f32->unit->f16 vs fused)
    assert r2[1, 0] == 17 and r2[3, 0] == 333 and s2[1, 0] > 0.999
    gal.close()


def _blur_crops():
    from frhot import synth
    rng = np.random.default_rng(21)
    shapes = [(1, 1), (1, 2), (2, 1), (2, 2), (1, 300), (300, 1), (31, 63), (32, 64), (33, 65), (64, 128), (65, 129),
              (7, 1000), (1000, 7), (224, 224), (448, 448), (601, 97), (130, 517), (1500, 1400)]
    imgs = [synth.crop(rng, h, w) for h, w in shapes]
    imgs.append(rng.integers(0, 256, (200, 333, 3), dtype=np.uint8))    # white noise: the largest sums per pixel
    imgs.append(np.zeros((50, 70, 3), np.uint8))
    imgs.append(np.full((50, 70, 3), 255, np.uint8))
    chk = np.zeros((96, 96, 3), np.uint8)
    chk[::2, ::2] = 255
    chk[1::2, 1::2] = 255
    imgs.append(chk)                                                    # every Laplacian value is +-1020
    base = synth.crop(rng, 400, 500)
    imgs.append(base[13:390:2, 5:480])                                  # strided view: every other row
    imgs.append(base[:, 100:164])                                       # a column band of a wider image
    return imgs


def test_crop_sharpness_exact(gpu_ctx):
    """The blur gate's measure (legacy.py:125-129): integer sums bit-exact against the oracle, the
    variance against numpy's .var() of the oracle's float64 Laplacian, and the gate's decision."""
    import os
    from oracle import quality as oq
    imgs = _blur_crops()
    sums = gpu_ctx.crop_sharpness_sums(imgs)
    var = gpu_ctx.crop_sharpness(imgs)
    for i, im in enumerate(imgs):
        assert tuple(int(v) for v in sums[i]) == oq.laplacian_sums(im), f"crop {i} {im.shape}"
        want = oq.laplacian_var(im)
        assert var[i] == pytest.approx(want, rel=1e-11, abs=1e-9), f"crop {i} {im.shape}"
        assert bool(var[i] >= 100) == oq.check_detect_blur(im)
    # one at a time, and in another order, the answers are the same (sums are accumulated by atomics)
    order = np.random.default_rng(0).permutation(len(imgs))
    assert np.array_equal(gpu_ctx.crop_sharpness_sums([imgs[i] for i in order]), sums[order])
    assert np.array_equal(gpu_ctx.crop_sharpness_sums([imgs[5]]), sums[5:6])
    # the committed fixture and the hand-worked cases, on the device
    import json
    golden = os.path.join(os.path.dirname(__file__), "golden")
    z = np.load(os.path.join(golden, "blur_oracle.npz"))
    got = gpu_ctx.crop_sharpness_sums([z[f"crop{i}"] for i in range(5)])
    assert np.array_equal(got, np.stack([z[f"sums{i}"] for i in range(5)]))
    with open(os.path.join(golden, "blur_hand_cases.json")) as f:
        for case in json.load(f):
            img = np.repeat(np.array(case["grey"], np.uint8)[:, :, None], 3, axis=2)
            assert gpu_ctx.crop_sharpness_sums([img]).tolist() == [[case["sum"], case["sumsq"]]]


def test_crop_sharpness_dev_and_gated_recognise(gpu_ctx):
    """The device-pointer form and the fused call give the host call's sums; the fused call's other
    answers are frhot_recognise's."""
    import ctypes as C
    import torch
    from frhot import synth
    from frhot._lib import check, load
    from frhot.engine import Gallery
    imgs = synth.crops(seed=41, n=6, lo=90, hi=260)
    want = gpu_ctx.crop_sharpness_sums(imgs)
    packed, offsets, hw, strides = synth.pack_crops(imgs)
    dev = torch.from_numpy(packed).cuda()
    out = torch.full((len(imgs), 2), -7, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    offsets = np.ascontiguousarray(offsets, np.int64)
    hw = np.ascontiguousarray(hw, np.int32)
    strides = np.ascontiguousarray(strides, np.int32)
    check(load().frhot_crop_sharpness_dev(
        gpu_ctx.handle, C.c_void_p(dev.data_ptr()), offsets.ctypes.data_as(C.POINTER(C.c_int64)),
        hw.ctypes.data_as(C.POINTER(C.c_int32)), strides.ctypes.data_as(C.POINTER(C.c_int32)), len(imgs),
        C.c_void_p(out.data_ptr())))
    gpu_ctx.sync()
    assert np.array_equal(out.cpu().numpy(), want)
    g = synth.unit_gallery(seed=42, n=500, d=4096, sparse=True)
    gal = Gallery(gpu_ctx, 4096, 500)
    gal.upsert(g, 0, np.arange(500))
    for gamma in (False, True):
        s1, r1, e1 = gpu_ctx.recognise(gal, imgs, None, 5, gamma=gamma, want_embeddings=True)
        s2, r2, e2, var = gpu_ctx.recognise(gal, imgs, None, 5, gamma=gamma, want_embeddings=True, want_sharpness=True)
        assert np.array_equal(s1, s2) and np.array_equal(r1, r2) and np.array_equal(e1, e2)
        assert np.array_equal(var, gpu_ctx.crop_sharpness(imgs))       # measured before the LUT either way
    gal.close()


def test_concurrent_host_calls_overlap_safely(gpu_ctx):
    """Host calls from several threads at once (the reference embeds from up to 10 worker threads,
    face_service.py:43,231-233): each goes through one of two staging slots, packing and upload
    of one call overlapping the compute of another.  Every caller gets exactly the answer it gets
    alone, for crops of very different sizes (the slots' buffers grow) and for both entry points."""
    from concurrent.futures import ThreadPoolExecutor
    from frhot import synth
    from frhot.engine import Gallery
    rng = np.random.default_rng(90)
    batches = []
    for i in range(6):
        n = int(rng.integers(1, 17))
        lo, hi = ((40, 90), (300, 700), (100, 200))[i % 3]
        batches.append(synth.crops(seed=900 + i, n=n, lo=lo, hi=hi))
    g = synth.unit_gallery(seed=91, n=2000, d=4096, sparse=True)
    gal = Gallery(gpu_ctx, 4096, 2000)
    gal.upsert(g, 0, np.arange(2000))
    alone_e = [gpu_ctx.embed(b, gamma=True) for b in batches]
    alone_r = [gpu_ctx.recognise(gal, b, None, 5, gamma=True, want_embeddings=True) for b in batches]

    def work(j):
        i = j % len(batches)
        if j % 2:
            return ("e", i, gpu_ctx.embed(batches[i], gamma=True))
        return ("r", i, gpu_ctx.recognise(gal, batches[i], None, 5, gamma=True, want_embeddings=True))

    with ThreadPoolExecutor(max_workers=4) as pool:
        results = list(pool.map(work, range(48)))
    for kind, i, out in results:
        if kind == "e":
            assert np.array_equal(out, alone_e[i])
        else:
            assert np.array_equal(out[0], alone_r[i][0]) and np.array_equal(out[1], alone_r[i][1])
            assert np.array_equal(out[2], alone_r[i][2]) and np.array_equal(out[2], alone_e[i])
    gal.close()


def test_errors_are_reported(gpu_ctx):
    from frhot import FrhotError
    from frhot.engine import Gallery
    with pytest.raises(FrhotError):
        gpu_ctx.embed([np.zeros((10, 4000, 3), np.uint8)])          # resizes to zero rows
    with pytest.raises(FrhotError):
        Gallery(gpu_ctx, 100, 10)                                    # D not a multiple of 64
    gal = Gallery(gpu_ctx, 64, 10)
    with pytest.raises(FrhotError):
        gal.upsert(np.ones((1, 64), np.float32), 0, [10])            # row beyond capacity
    with pytest.raises(FrhotError):
        gal.search(np.ones((1, 64), np.float32), None, 17)           # k too large
    s, r = gal.search(np.ones((2, 64), np.float32), None, 5)         # empty gallery
    assert (r == -1).all() and np.isneginf(s).all()
    gal.close()
    # the blur measure: a crop the embedding refuses (it resizes to nothing) is measured all the same;
    # bad arguments are refused with the library's codes
    import ctypes as C
    from frhot._lib import ECAPACITY, EINVAL, load
    wide = np.zeros((10, 4000, 3), np.uint8)
    wide[:, ::2] = 200
    from oracle import quality as oq
    assert gpu_ctx.crop_sharpness_sums([wide]).tolist() == [list(oq.laplacian_sums(wide))]
    lib = load()
    hw = np.array([[10, 4000]], np.int32)
    ptr = (C.POINTER(C.c_uint8) * 1)(wide.ctypes.data_as(C.POINTER(C.c_uint8)))
    out = np.zeros(2, np.int64)
    i32, i64 = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    bad_stride = np.array([100], np.int32)          # fewer bytes per row than 3 * w
    assert lib.frhot_crop_sharpness(gpu_ctx.handle, ptr, hw.ctypes.data_as(i32), bad_stride.ctypes.data_as(i32), 1,
                                    out.ctypes.data_as(i64)) == EINVAL
    ok_stride = np.array([12000], np.int32)
    assert lib.frhot_crop_sharpness(gpu_ctx.handle, ptr, hw.ctypes.data_as(i32), ok_stride.ctypes.data_as(i32),
                                    gpu_ctx.max_batch + 1, out.ctypes.data_as(i64)) == ECAPACITY
    assert lib.frhot_crop_sharpness(gpu_ctx.handle, ptr, hw.ctypes.data_as(i32), ok_stride.ctypes.data_as(i32), 1, None) == EINVAL
    assert b"null" in lib.frhot_last_error()


def test_row_sharded_search_dev_merge(gpu_ctx):
    """The N > 1 data path on one GPU: two row shards searched with frhot_search_dev (row_base
    added), lists stacked as an all-gather would, merged with frhot_merge_topk_dev; the result
    must equal one search over the whole gallery, including a tie that straddles the shards."""
    import torch
    from frhot import synth
    from frhot.dist import ShardedSearch, shard_rows
    from frhot.engine import Gallery
    from oracle import search as osr
    n, d, bq, k, P = 6001, 512, 37, 5, 2
    g = synth.unit_gallery(seed=9, n=n, d=d)
    g[n - 1] = g[0]
    rng = np.random.default_rng(10)
    targets = rng.integers(0, n, size=bq)
    q = synth.planted_queries(rng, g[targets], 0.8)
    q[0] = g[0]
    dev = torch.device("cuda", 0)
    q16 = torch.from_numpy(osr.unit_rows(q).astype(np.float16)).to(dev)
    shards, lists_s, lists_r = [], [], []
    for r in range(P):
        base, rows = shard_rows(n, P, r)
        gal = Gallery(gpu_ctx, d, rows)
        gal.upsert(g[base:base + rows], 0, np.arange(rows))
        s_out = torch.empty((bq, k), dtype=torch.float32, device=dev)
        r_out = torch.empty((bq, k), dtype=torch.int64, device=dev)
        gal.search_dev(q16.data_ptr(), None, bq, k, s_out.data_ptr(), r_out.data_ptr(), base)
        shards.append(gal)
        lists_s.append(s_out)
        lists_r.append(r_out)
    gpu_ctx.sync()                      # the library enqueues on its own stream
    s_all = torch.stack(lists_s).contiguous()
    r_all = torch.stack(lists_r).contiguous()
    torch.cuda.synchronize()
    out_s = torch.empty((bq, k), dtype=torch.float32, device=dev)
    out_r = torch.empty((bq, k), dtype=torch.int64, device=dev)
    gpu_ctx.merge_topk_dev(s_all.data_ptr(), r_all.data_ptr(), P, bq, k, out_s.data_ptr(), out_r.data_ptr())
    gpu_ctx.sync()
    got_s, got_r = out_s.cpu().numpy(), out_r.cpu().numpy()
    ref_s, ref_r = osr.cosine_topk(q, g, k, gallery_dtype=np.float16)
    full = osr.unit_rows(q).astype(np.float16).astype(np.float32) @ osr.unit_rows(g).astype(np.float16).astype(np.float32).T
    _check_topk(got_s, got_r, ref_s, ref_r, full, tol=2e-5)
    assert list(got_r[0, :2]) == [0, n - 1]
    # the same lists as ONE all-gather leaves them: per shard a block of rows then scores (37 x 5 is
    # odd, so a block ends in four bytes of padding), merged in place, must give the same bits
    nk = bq * k
    words = nk + (nk + 1) // 2
    packed = torch.full((P, words), -7, dtype=torch.int64, device=dev)
    for r in range(P):
        packed[r, :nk] = lists_r[r].view(-1)
        packed[r, nk:].view(torch.float32)[:nk] = lists_s[r].view(-1)
    out_s2 = torch.empty_like(out_s)
    out_r2 = torch.empty_like(out_r)
    torch.cuda.synchronize()
    gpu_ctx.merge_topk_packed_dev(packed.data_ptr(), words * 8, P, bq, k, out_s2.data_ptr(), out_r2.data_ptr())
    gpu_ctx.sync()
    assert np.array_equal(out_r2.cpu().numpy(), got_r) and np.array_equal(out_s2.cpu().numpy(), got_s)
    lib = gpu_ctx._lib
    import ctypes as C
    args = (P, bq, k, C.c_void_p(out_s2.data_ptr()), C.c_void_p(out_r2.data_ptr()))
    assert lib.frhot_merge_topk_packed_dev(gpu_ctx.handle, C.c_void_p(packed.data_ptr()), nk * 12 - 4, *args) == -1
    assert lib.frhot_merge_topk_packed_dev(gpu_ctx.handle, C.c_void_p(packed.data_ptr()), words * 8 + 4, *args) == -1
    for gal in shards:
        gal.close()


def test_full_size_scan_properties(gpu_ctx):
    """The benchmark's gallery (10 M x 512 fp16, 256 queries, top-5), checked through properties
    that need no CPU pass over 10 GB: a stored row asked for comes back first with score 1 (its
    planted duplicate second, lower row first); scores descend; every returned score is the dot
    product of the row it names; a permutation of the queries permutes the answers; the k-th score is not beaten by any row of a random sample."""
    from frhot.engine import Gallery
    n, d, bq, k = 10_000_000, 512, 256, 5
    gal = Gallery(gpu_ctx, d, n)
    gal.fill_synthetic(seed=0xFACE0100, row0=0, n=n, nkeys=0, dup_every=100)   # row 100j+99 copies 100j+98
    rng = np.random.default_rng(5)
    asked = rng.integers(0, n, size=bq)
    asked[:8] = [0, 98, 99, n - 1, n - 2, 255, 256, 9_999_898]
    rows16, _ = gal.read(asked)
    q = rows16.astype(np.float32)
    s, r = gal.search(q, None, k)
    assert (np.diff(s, axis=1) <= 0).all() and (r >= 0).all()
    twin = np.where(asked % 100 == 98, asked + 1, np.where(asked % 100 == 99, asked - 1, -1))
    for i in range(bq):
        if twin[i] < 0:
            assert r[i, 0] == asked[i]
        else:
            assert list(r[i, :2]) == sorted([asked[i], twin[i]])
    norms = (rows16.astype(np.float64) ** 2).sum(1)
    assert np.abs(s[:, 0] - norms).max() <= 2e-3               # fp16 unit rows: |row|^2 within rounding of 1
    back, _ = gal.read(r.reshape(-1))
    q16 = q / np.sqrt((q.astype(np.float64) ** 2).sum(1, keepdims=True)).astype(np.float32)
    q16 = q16.astype(np.float16).astype(np.float32)
    redo = (back.astype(np.float32).reshape(bq, k, d) * q16[:, None, :]).sum(-1)
    assert np.abs(redo - s).max() <= 2e-5
    perm = rng.permutation(bq)
    s2, r2 = gal.search(q[perm], None, k)
    assert np.array_equal(r2, r[perm]) and np.array_equal(s2, s[perm])
    probe = rng.integers(0, n, size=20000)
    pr, _ = gal.read(probe)
    ps = q16[:32] @ pr.astype(np.float32).T
    assert (ps.max(axis=1) <= s[:32, 0] + 2e-5).all()
    for i in range(32):
        beat = probe[ps[i] > s[i, k - 1] + 2e-5]
        assert np.isin(beat, r[i]).all()
    gal.close()


def test_gallery_write_f16_restores_bit_for_bit(gpu_ctx, tmp_path):
    """frhot_gallery_read -> frhot_gallery_write_f16 round trip, and a service snapshot restored
    into a fresh device gallery answers every search identically."""
    from frhot import synth
    from frhot.engine import Gallery
    from frhot.gallery_service import GalleryService
    n, d = 300, 4096
    g = synth.unit_gallery(seed=41, n=n, d=d, sparse=True)
    gal = Gallery(gpu_ctx, d, n)
    keys = (np.arange(n) % 3).astype(np.int32)
    gal.upsert(g, keys, np.arange(n))
    rows16, k16 = gal.read(np.arange(n))
    gal2 = Gallery(gpu_ctx, d, n)
    perm = np.random.default_rng(0).permutation(n)
    gal2.write_f16(rows16[perm], k16[perm], perm)
    back, kback = gal2.read(np.arange(n))
    assert np.array_equal(back.view(np.uint16), rows16.view(np.uint16)) and np.array_equal(kback, keys)
    q = g[:20] + 0.01
    s1, r1 = gal.search(q, keys[:20], 5)
    s2, r2 = gal2.search(q, keys[:20], 5)
    assert np.array_equal(r1, r2) and np.array_equal(s1, s2)
    gal.close()
    gal2.close()
    svc = GalleryService(gpu_ctx, capacity=64)
    svc.create_collection("S9_Employees")
    for i in range(12):
        assert svc.insert_point("S9_Employees", list(g[i]), f"p{i % 4}", f"n{i % 4}", "S9", time_created=f"t{i:02d}")[0] == 201
    before = [svc.search_point("S9_Employees", list(g[i]), "S9") for i in range(12)]
    st, info = svc.create_snapshot("S9_Employees", str(tmp_path))
    assert st == 200 and info["points"] == 12
    svc2 = GalleryService(gpu_ctx, capacity=64)
    assert svc2.recover_snapshot("S9_Employees", str(tmp_path / "S9_Employees" / info["name"]))[0] == 200
    assert [svc2.search_point("S9_Employees", list(g[i]), "S9") for i in range(12)] == before
    svc.close()
    svc2.close()


def test_service_enrols_and_deletes_while_recognitions_are_in_flight(gpu_ctx):
    """qdrant_database_FE serves inserts, deletes and searches concurrently (app.py:242-438); the
    in-process service gives up its table lock while the device works on a recognition, so
    enrolments and deletions (whose rows are used again) run between and beside the scans.  Every
    recognition still names the person it asked for, and the table ends consistent."""
    import threading
    from frhot import synth
    from frhot.gallery_service import GalleryService
    svc = GalleryService(gpu_ctx, capacity=1024)
    col = "S_Employees"
    assert svc.create_collection(col)[0] == 201
    stores = ["s0", "s1", "s2", "s3"]
    crops = synth.crops(seed=123, n=len(stores), lo=120, hi=260)
    emb = gpu_ctx.embed(crops, gamma=True)
    for j, st in enumerate(stores):
        assert svc.insert_point(col, list(emb[j]), f"fixed{j}", f"F{j}", st)[0] == 201
    stop = threading.Event()
    errors = []
    counts = {"recognitions": 0, "mutations": 0}

    def mutate():
        rng = np.random.default_rng(5)
        live = []
        try:
            i = 0
            while not stop.is_set():
                v = rng.standard_normal(4096).astype(np.float32)
                st = stores[i % len(stores)]
                assert svc.insert_point(col, list(v), f"temp{i}", f"T{i}", st)[0] == 201
                live.append(i)
                if len(live) > 3:
                    assert svc.delete_point(col, f"temp{live.pop(0)}")[0] == 200
                counts["mutations"] += 1
                i += 1
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    def recognise(seed):
        rng = np.random.default_rng(seed)
        try:
            while not stop.is_set():
                pick = [int(x) for x in rng.integers(0, len(stores), size=int(rng.integers(1, 5)))]
                out = svc.recognise_points([(col, crops[j], stores[j]) for j in pick], gamma=True)
                for j, (status, content, _e) in zip(pick, out):
                    assert status == 200 and content["message"] == "Point found", content
                    score, payload = content["data"][0]
                    assert payload["id"] == f"fixed{j}" and score > 0.999, (j, score, payload)
                counts["recognitions"] += len(pick)
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=mutate)] + [threading.Thread(target=recognise, args=(s,)) for s in (1, 2, 3)]
    for t in threads:
        t.start()
    import time
    time.sleep(1.5)
    stop.set()
    for t in threads:
        t.join(30)
    assert not any(t.is_alive() for t in threads)
    assert not errors, errors[:3]
    assert counts["recognitions"] > 20 and counts["mutations"] > 20, counts
    assert svc.point_count(col) == len(stores) + 3
    for j, st in enumerate(stores):
        status, content = svc.search_point(col, list(emb[j]), st)
        assert status == 200 and content["data"][0][1]["id"] == f"fixed{j}"
    svc.close()


@pytest.mark.parametrize("fused", [False, True])
def test_api_end_to_end_concurrent_requests(gpu_ctx, fused):
    """The two routes served from the GPU context (frhot.api.create_app): enrol, then recognise
    from many requests in flight at once.  The behaviour the reference's manual scripts exercise
    (test/register.py then test/detect.py: the enrolled picture is recognised as its id and name),
    plus the grouping of concurrent embeds and searches into shared device calls."""
    import asyncio
    import base64
    import io

    import httpx
    from PIL import Image

    from frhot import synth
    from frhot.api import create_app

    def b64(img):
        buf = io.BytesIO()
        Image.fromarray(img[:, :, ::-1]).save(buf, format="PNG")
        return base64.b64encode(buf.getvalue()).decode()

    rng = np.random.default_rng(77)
    people = [b64(synth.crop(rng, int(rng.integers(120, 300)), int(rng.integers(120, 300)))) for _ in range(12)]
    app = create_app(gpu_ctx, capacity=256, api_key="k", max_delay_s=0.05, fused=fused)
    H = {"X-API-Key": "k"}

    async def run():
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://frhot") as c:
            # is_update skips the "face exists" pre-search: descriptors of a randomly initialised
            # network are all close to each other, so a second person would be answered 409
            for i, img in enumerate(people):
                r = await c.post("/create_face_img_base64", headers=H, json={
                    "img_base64": img, "id": str(100 + i), "name": f"P{i}", "role": "1", "store_id": "S", "is_update": True})
                assert r.status_code == 201, r.text
            r = await c.post("/create_face_img_base64", headers=H, json={
                "img_base64": people[0], "id": "999", "name": "Dup", "role": "1", "store_id": "S"})
            assert r.status_code == 409
            b0, s0 = app.state.batcher.batches, app.state.search_batcher.batches
            f0 = app.state.recogniser.batches if fused else 0
            rs = await asyncio.gather(*[c.post("/face_recog_img_base64", headers=H, json={
                "img_base64": img, "role": "1", "store_id": "S"}) for img in people])
            for i, r in enumerate(rs):
                assert r.status_code == 200 and r.json() == {"status": 1, "id": str(100 + i), "name": f"P{i}"}, r.text
            if fused:
                assert app.state.recogniser.batches - f0 < len(people)         # requests shared fused device calls
                assert app.state.batcher.batches == b0 and app.state.search_batcher.batches == s0
            else:
                assert app.state.batcher.batches - b0 < len(people)            # crops shared device calls
                assert app.state.search_batcher.batches - s0 < len(people)     # and so did the searches
            r = await c.post("/face_recog_img_base64", headers=H, json={"img_base64": people[0], "role": "0", "store_id": "S"})
            assert r.status_code == 404                                    # nothing enrolled among the customers
            r = await c.post("/face_recog_img_base64", json={"img_base64": people[0], "role": "1", "store_id": "S"})
            assert r.status_code == 403
    try:
        asyncio.run(run())
    finally:
        app.state.close()


@pytest.mark.parametrize("fused", [False, True])
def test_api_check_in_blur_gate_on_the_device(gpu_ctx, fused):
    """face_service.py:193,212-219 through the routes: a sharp photo of an enrolled employee is
    recognised, the same photo blurred is refused with the reference's body; the measure comes from
    the device (with `fused` from the same call that embeds and searches), and customers
    (is_checkin False) are not gated."""
    import asyncio
    import base64
    import io

    import httpx
    from PIL import Image, ImageFilter

    from frhot import synth
    from frhot.api import create_app
    from oracle import quality as oq

    rng = np.random.default_rng(78)
    sharp = synth.crop(rng, 240, 200)
    sharp[::3, ::2] = 255 - sharp[::3, ::2]
    blurred = np.asarray(Image.fromarray(sharp).filter(ImageFilter.GaussianBlur(4)))
    assert oq.laplacian_var(sharp) > 100 > oq.laplacian_var(blurred)

    def b64(img):
        buf = io.BytesIO()
        Image.fromarray(img[:, :, ::-1]).save(buf, format="PNG")
        return base64.b64encode(buf.getvalue()).decode()

    app = create_app(gpu_ctx, capacity=256, api_key="k", max_delay_s=0.01, fused=fused, quality_gates=True)
    H = {"X-API-Key": "k"}

    async def run():
        async with httpx.AsyncClient(transport=httpx.ASGITransport(app=app), base_url="http://frhot") as c:
            for role in ("1", "0"):
                r = await c.post("/create_face_img_base64", headers=H, json={
                    "img_base64": b64(sharp), "id": "5", "name": "Eve", "role": role, "store_id": "S"})
                assert r.status_code == 201, r.text
            rs = await asyncio.gather(*[c.post("/face_recog_img_base64", headers=H, json={
                "img_base64": b64(img), "role": role, "store_id": "S"})
                for img, role in ((sharp, "1"), (blurred, "1"), (blurred, "0"), (sharp, "1"))])
            assert rs[0].status_code == 200 and rs[0].json() == {"status": 1, "id": "5", "name": "Eve"}, rs[0].text
            assert rs[1].status_code == 400 and rs[1].json() == {
                "status": 2, "message": "Face is blur! Please keep your face in focus"}, rs[1].text
            assert rs[2].status_code in (200, 404)          # not gated; whether the blurred photo still matches is the network's affair
            assert rs[3].status_code == 200
    try:
        asyncio.run(run())
    finally:
        app.state.close()


def test_sharded_search_exchange_path_on_rccl(gpu_ctx):
    """The multi-rank step (all-gather of queries, shard scan with a row base, all-gathers of the
    shard lists, merge) through a one-rank RCCL group on the library's stream: what one GPU can
    exercise of the N > 1 path.  Equals a plain search of the same gallery."""
    import os
    import socket

    import torch
    import torch.distributed as dist

    from frhot import synth
    from frhot.dist import ShardedSearch
    from frhot.engine import Gallery
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n, d, b, k, base = 50_000, 512, 23, 5, 1_000_000        # 23 x 5 entries: an odd count, blocks end in padding
    g = synth.unit_gallery(seed=4, n=n, d=d)
    q = synth.planted_queries(np.random.default_rng(2), g[:b], 0.8)
    gal = Gallery(gpu_ctx, d, n)
    gal.upsert(g, 0, np.arange(n))
    want_s, want_r = gal.search(q, None, k)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        q16 = torch.from_numpy(q.astype(np.float16)).to(dev)

        def search_local(q_all, s_out, r_out):
            gal.search_dev(q_all.data_ptr(), None, q_all.shape[0], k, s_out.data_ptr(), r_out.data_ptr(), base)

        def merge(s_all, r_all, out_s, out_r):
            assert s_all.data_ptr() == r_all.data_ptr() + b * k * 8
            gpu_ctx.merge_topk_packed_dev(r_all.data_ptr(), sharded.block_bytes, 1, b, k, out_s.data_ptr(),
                                          out_r.data_ptr())

        sharded = ShardedSearch(1, b, d, k, dev, search_local, merge, always_exchange=True)
        torch.cuda.synchronize()
        out_s, out_r = sharded(q16)
        gpu_ctx.sync()
        torch.cuda.synchronize()
        got_s, got_r = out_s.cpu().numpy(), out_r.cpu().numpy()
    finally:
        dist.destroy_process_group()
        gal.close()
    assert np.array_equal(got_r, want_r + base)
    assert np.abs(got_s - want_s).max() <= 2e-3        # the host call normalises float32 queries; here they arrive as fp16


@pytest.mark.parametrize("with_keys", [False, True])
def test_search_sharded_dev_inside_the_library(gpu_ctx, with_keys):
    """frhot_search_sharded_dev (RCCL bound by the library itself, no torch.distributed) in a
    communicator of one rank: both all-gathers, the shard scan with a row base and the merge run;
    the answer equals frhot_search_dev's on the same fp16 queries bit for bit.  Twice in a row, the
    second call with fewer queries (the exchange buffers are reused)."""
    import torch

    from frhot import synth
    from frhot.engine import Context, Gallery
    dev = torch.device("cuda", 0)
    n, d, k, base = 60_000, 512, 5, 3_000_000
    g = synth.unit_gallery(seed=14, n=n, d=d)
    keys = (np.arange(n) % 37).astype(np.int32)
    gal = Gallery(gpu_ctx, d, n)
    gal.upsert(g, keys, np.arange(n))
    gpu_ctx.dist_init(0, 1, Context.dist_unique_id())
    try:
        with pytest.raises(Exception):
            gpu_ctx.dist_init(0, 1, Context.dist_unique_id())       # joined already
        for b in (23, 7):
            q = synth.planted_queries(np.random.default_rng(b), g[100:100 + b], 0.8)
            q16 = torch.from_numpy(q.astype(np.float16)).to(dev)
            qk = torch.from_numpy(keys[100:100 + b].copy()).to(dev) if with_keys else None
            want_s = torch.empty((b, k), dtype=torch.float32, device=dev)
            want_r = torch.empty((b, k), dtype=torch.int64, device=dev)
            got_s, got_r = torch.full_like(want_s, 7.0), torch.full_like(want_r, -7)
            torch.cuda.synchronize()
            gal.search_dev(q16.data_ptr(), qk.data_ptr() if with_keys else None, b, k, want_s.data_ptr(),
                           want_r.data_ptr(), base)
            gal.search_sharded_dev(q16.data_ptr(), qk.data_ptr() if with_keys else None, b, k, got_s.data_ptr(),
                                   got_r.data_ptr(), base)
            gpu_ctx.sync()
            assert torch.equal(got_r, want_r) and torch.equal(got_s, want_s)
            assert int(got_r[:, 0].min()) >= base
            if with_keys:
                assert np.array_equal((got_r.cpu().numpy() - base) % 37, np.repeat(keys[100:100 + b, None], k, 1))
    finally:
        gpu_ctx.dist_shutdown()
        gal.close()
    with pytest.raises(Exception):
        gal2 = Gallery(gpu_ctx, d, 16)
        try:
            gal2.search_sharded_dev(1, None, 1, 1, 1, 1)        # no communicator any more: refused before any use
        finally:
            gal2.close()


def test_sharded_service_one_rank_equals_single_service(gpu_ctx):
    """ShardedGalleryService over the HIP backend with ONE rank (no process group: the collectives
    fall away, everything else runs: global-row placement, the device-pointer embed, the search by key
    with a row base) against GalleryService on the same GPU: enrolments, an eviction run past ten
    faces of one id, a deletion, and fused recognitions of crops answer identically."""
    import torch

    from frhot import synth
    from frhot.engine import Gallery
    from frhot.gallery_service import GalleryService
    from frhot.sharded_service import HipShardBackend, ShardedGalleryService
    dev = torch.device("cuda", 0)
    cap = 2048
    single = GalleryService(gpu_ctx, capacity=cap)
    svc = ShardedGalleryService(HipShardBackend(gpu_ctx, Gallery(gpu_ctx, 4096, cap), 0, cap, dev), 0, 1, cap, 4096,
                                batch=16)
    crops = synth.crops(seed=91, n=14, lo=120, hi=260)
    emb = gpu_ctx.embed(crops, gamma=True)
    for s in (single, svc):
        assert s.create_collection("S_Employees")[0] == 201 and s.create_collection("S_Customers")[0] == 201
    ops = []
    for i in range(6):
        ops.append(("insert_point", dict(collection_name="S_Employees", vector_embedding=emb[i].tolist(), id=str(i),
                                         name=f"n{i}", store_id="S", time_created=f"2026/01/01 00:00:{i:02d}")))
    for j in range(12):         # twelve faces of one id: the limit of ten evicts the oldest
        ops.append(("insert_point", dict(collection_name="S_Customers", vector_embedding=emb[6 + j % 4].tolist(), id="77",
                                         name="many", store_id="S", time_created=f"2026/01/01 00:01:{j:02d}")))
    ops.append(("delete_point", "S_Employees", "3"))
    got = svc.mutate(ops)
    want = [single.insert_point(**op[1]) if op[0] == "insert_point" else single.delete_point(op[1], op[2]) for op in ops]
    assert got == want
    assert svc.point_count("S_Customers") == single.point_count("S_Customers") == 10
    reqs = [("S_Employees", crops[i], "S") for i in range(6)] + [("S_Customers", crops[7], "S"), ("S_Employees", crops[12], "S"),
                                                                  ("nowhere", crops[0], "S"), ("S_Employees", crops[1], "T")]
    a = single.recognise_points(reqs, gamma=True)
    b = svc.recognise_points(reqs, gamma=True)
    assert len(a) == len(b) == len(reqs)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x[0] == y[0] and x[1]["message"] == y[1]["message"], (i, x, y)
        if x[1].get("data"):
            assert x[1]["data"][0][1] == y[1]["data"][0][1] and abs(x[1]["data"][0][0] - y[1]["data"][0][0]) < 1e-6
    assert a[0][1]["data"][0][1]["id"] == "0" and a[8][0] == 404 and a[9][1]["data"] == []
    sv = [(c, emb[i].tolist(), s) for i, (c, _crop, s) in enumerate(reqs[:6])]
    for x, y in zip(single.search_points(sv), svc.search_points(sv)):
        assert x[0] == y[0] and x[1]["message"] == y[1]["message"]
        if x[1].get("data"):
            assert x[1]["data"][0][1] == y[1]["data"][0][1] and abs(x[1]["data"][0][0] - y[1]["data"][0][0]) < 1e-6
    svc.close()
    single.close()
