"""GPU parity at the sizes of the benchmark configurations, through the C ABI, plus the committed
golden fixtures replayed on the HIP path.

The large galleries are filled on the device (frhot_gallery_fill_synthetic); the CPU side reads
the stored fp16 rows back and the oracle scores them as they stand (oracle.search.topk_stored),
so the comparison is exact up to float32 summation order: identical row lists except inside
score ties within 2e-5."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _read_all(gal, n, step=100_000):
    out = np.empty((n, gal.dim), dtype=np.float16)
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        out[lo:hi], _ = gal.read(np.arange(lo, hi))
    return out


def _check_lists(got_s, got_r, ref_s, ref_r, score, tol=2e-5):
    """Identical row lists except inside score ties within tol; scores within tol."""
    assert got_r.shape == ref_r.shape
    for i in range(ref_r.shape[0]):
        if np.array_equal(got_r[i], ref_r[i]):
            continue
        for j in range(ref_r.shape[1]):
            if got_r[i, j] == ref_r[i, j]:
                continue
            assert (got_r[i, j] >= 0) == (ref_r[i, j] >= 0), f"query {i} slot {j}: presence differs"
            assert abs(score(i, int(got_r[i, j])) - ref_s[i, j]) <= tol, \
                f"query {i} slot {j}: row {got_r[i, j]} vs {ref_r[i, j]} is not a tie"
    ok = np.isfinite(ref_s)
    assert np.array_equal(np.isfinite(got_s), ok)
    if ok.any():
        assert np.abs(got_s[ok] - ref_s[ok]).max() <= tol


def _noisy_queries(rng, rows16, cos):
    """fp16 unit queries near the given stored rows."""
    from frhot import synth
    q = synth.planted_queries(rng, rows16.astype(np.float32), cos)
    return q.astype(np.float16)


def _search_f16(gal, q16, qkeys, k, static=False):
    """frhot_search_dev(_static) with fp16 queries exactly as given (the host form would re-normalise)."""
    import torch
    dev = torch.device("cuda", 0)
    bq = q16.shape[0]
    qd = torch.from_numpy(np.ascontiguousarray(q16)).to(dev)
    kd = None if qkeys is None else torch.from_numpy(np.ascontiguousarray(qkeys, dtype=np.int32)).to(dev)
    s = torch.empty((bq, k), dtype=torch.float32, device=dev)
    r = torch.empty((bq, k), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    fn = gal.search_dev_static if static else gal.search_dev
    fn(qd.data_ptr(), None if kd is None else kd.data_ptr(), bq, k, s.data_ptr(), r.data_ptr(), 0)
    gal.ctx.sync()
    return s.cpu().numpy(), r.cpu().numpy()


def test_golden_fixtures_on_the_hip_path(gpu_ctx):
    """tests/golden/preprocess_oracle.npz and search_oracle.npz through the library."""
    from frhot.engine import Gallery
    z = np.load(os.path.join(GOLDEN, "preprocess_oracle.npz"))
    crops = [z[f"crop{i}"] for i in range(3)]
    gpu_ctx.embed(crops, gamma=True, flip=True)
    u8 = gpu_ctx.debug_read(0, 3)
    for i in range(3):
        assert np.array_equal(u8[i], z[f"u8_{i}"]), f"golden crop {i}"
    lut = np.load(os.path.join(GOLDEN, "gamma_lut_1p5.npy"))
    ramp = np.repeat(np.arange(256, dtype=np.uint8), 3).reshape(1, 256, 3)
    ramp = np.ascontiguousarray(np.broadcast_to(ramp, (224, 256, 3))[:, :224])     # identity resize of values 0..223
    gpu_ctx.embed([ramp], gamma=True, flip=False)
    assert np.array_equal(gpu_ctx.debug_read(0, 1)[0, 0, :, 0], lut[:224])
    z = np.load(os.path.join(GOLDEN, "search_oracle.npz"))
    g = z["g"].astype(np.float32)
    gal = Gallery(gpu_ctx, g.shape[1], g.shape[0])
    gal.upsert(g, z["keys"], np.arange(g.shape[0]))
    s, r = gal.search(z["q"], z["qkeys"], 5)
    assert np.array_equal(r, z["rows"])
    assert np.abs(s - z["scores"]).max() <= 1e-3        # the fixture's scores are float32-gallery scores
    gal.close()


def test_config2_full_size_scan_matches_oracle(gpu_ctx):
    """Gallery scan only: 1 M x 512 fp16, 64 queries, top-10, every answer against the oracle."""
    from frhot.engine import Gallery
    from oracle import search as osr
    n, d, bq, k = 1_000_000, 512, 64, 10
    gal = Gallery(gpu_ctx, d, n)
    gal.fill_synthetic(seed=0xFACE0002, row0=0, n=n, nkeys=0, dup_every=100)
    g16 = _read_all(gal, n)
    rng = np.random.default_rng(2)
    targets = rng.integers(0, n, size=bq)
    targets[:4] = [98, 99, n - 1, 0]
    q16 = _noisy_queries(rng, g16[targets], 0.8)
    q16[:8] = g16[targets[:8]]                                   # stored rows themselves: duplicates tie exactly
    for static in (False, True):
        got_s, got_r = _search_f16(gal, q16, None, k, static=static)
        ref_s, ref_r = osr.topk_stored(q16, g16, k)
        qf = q16.astype(np.float32)
        _check_lists(got_s, got_r, ref_s, ref_r, lambda i, row: float(qf[i] @ g16[row].astype(np.float32)))
        assert list(got_r[0, :2]) == [98, 99] and list(got_r[1, :2]) == [98, 99]
        assert (got_r[:, 0] == np.where(targets % 100 == 99, targets - 1, targets)).all()
    # the host-call form (float32 queries, normalised inside) agrees
    hs, hr = gal.search(q16.astype(np.float32), None, k)
    assert (hr[:, 0] == got_r[:, 0]).all()
    gal.close()


def test_config3_embed_256_and_match_1m_x_4096(weights_f32):
    """Embed + match at the reference's dimension: 256 crops through the network, top-5 against a
    1 M x 4096 fp16 gallery (8.2 GB) that holds the descriptors of some of them.  Every crop's
    planted row comes back first; the fused call equals embed + search; eight queries are checked
    against the oracle over the whole gallery; every returned score is the dot product of its row."""
    from frhot import synth
    from frhot.engine import Context, Gallery
    from frhot.weights import pack_blob
    from oracle import search as osr
    n, d, b, k = 1_000_000, 4096, 256, 5
    ctx = Context(device=0, max_batch=b)
    try:
        ctx.load_model(pack_blob(weights_f32), input_scale=1.0 / 16.0)
        imgs = synth.crops(seed=303, n=b, lo=112, hi=300)
        emb = ctx.embed(imgs, gamma=True)
        gal = Gallery(ctx, d, n)
        gal.fill_synthetic(seed=0xFACE0003, row0=0, n=n, nkeys=0, dup_every=100)
        rng = np.random.default_rng(3)
        planted = rng.choice(n, size=b // 2, replace=False)
        planted[:3] = [0, n - 1, 255]
        gal.upsert(emb[: b // 2], 0, planted)                    # the first half of the crops are enrolled
        s, r, e2 = ctx.recognise(gal, imgs, None, k, gamma=True, want_embeddings=True)
        assert np.array_equal(e2, emb)
        assert np.array_equal(r[: b // 2, 0], planted) and (s[: b // 2, 0] > 0.999).all()
        assert (np.diff(s, axis=1) <= 0).all() and (r >= 0).all()
        # the host form rounds the float32 query to fp16 after its own normalisation: near-equal
        # scores further down the lists may swap, the planted row and the scores may not
        s2, r2 = gal.search(emb, None, k)
        assert np.array_equal(r2[: b // 2, 0], planted) and np.abs(s2 - s).max() <= 2e-3
        # descriptors of this synthetic network resemble each other, rows of the synthetic gallery do
        # not resemble them: the enrolled half's descriptors fill the other ranks
        back, _ = gal.read(r.reshape(-1))
        q16 = emb.astype(np.float16)
        # the fused path rounds the unit descriptor to fp16 once
        redo = (back.astype(np.float32).reshape(b, k, d) * q16.astype(np.float32)[:, None, :]).sum(-1)
        assert np.abs(redo - s).max() <= 1e-3
        pick = np.array([0, 1, 2, 100, 127, 128, 200, 255])
        qs = q16[pick]
        best_s = np.full((len(pick), 0), -np.inf, np.float32)
        best_r = np.zeros((len(pick), 0), np.int64)
        for lo in range(0, n, 50_000):
            rows16, _ = gal.read(np.arange(lo, lo + 50_000))
            cs, cr = osr.topk_stored(qs, rows16, k, row0=lo)
            ms = np.concatenate([best_s, cs], axis=1)
            mr = np.concatenate([best_r, cr], axis=1)
            order = np.lexsort((mr, -ms), axis=1)[:, :k]
            best_s, best_r = np.take_along_axis(ms, order, axis=1), np.take_along_axis(mr, order, axis=1)
        ds, dr = _search_f16(gal, qs, None, k)

        def score(i, row):
            return float(qs[i].astype(np.float32) @ gal.read([row])[0][0].astype(np.float32))

        _check_lists(ds, dr, best_s, best_r, score, tol=1e-4)
        assert np.array_equal(dr, r[pick]) or np.abs(ds - s[pick]).max() <= 1e-3
        gal.close()
    finally:
        ctx.close()


@pytest.mark.parametrize("bq", [1024, 2048])
def test_config4_rank_shape_matches_oracle(gpu_ctx, bq):
    """One rank's share of the 10 M x 512 gallery over 8 GPUs (1.25 M rows) against all the queries
    of a step (batch 1024, and 2048 = 8 ranks x 256 crops), top-5, every answer against the
    oracle; then the same shard as rank 3 of 8 (row base added) merged with seven empty lists."""
    from frhot.engine import Gallery
    from oracle import search as osr
    n, d, k = 1_250_000, 512, 5
    gal = Gallery(gpu_ctx, d, n)
    gal.fill_synthetic(seed=0xFACE0004, row0=0, n=n, nkeys=0, dup_every=100)
    g16 = _read_all(gal, n)
    rng = np.random.default_rng(bq)
    targets = rng.integers(0, n, size=bq)
    q16 = _noisy_queries(rng, g16[targets], 0.75)
    q16[::64] = g16[targets[::64]]
    got_s, got_r = _search_f16(gal, q16, None, k)
    ref_s, ref_r = osr.topk_stored(q16, g16, k)
    qf = q16.astype(np.float32)
    _check_lists(got_s, got_r, ref_s, ref_r, lambda i, row: float(qf[i] @ g16[row].astype(np.float32)))
    st_s, st_r = _search_f16(gal, q16, None, k, static=True)
    assert np.array_equal(st_r, got_r) and np.array_equal(st_s, got_s)
    gal.close()


def test_config5_rank_shape_keyed_4096_queries(gpu_ctx):
    """One rank's share of the 100 M x 512 gallery over 8 GPUs (12.5 M rows, 12.8 GB) with a key
    (store / role collection) per row out of 2,000, tombstones, and 4,096 keyed queries of top-5:
    every keyed query against the oracle over ALL rows of its key; ANY-key queries and a key
    nobody has through properties."""
    from frhot.engine import Gallery
    from oracle import search as osr, synth_gallery as sg
    n, d, bq, k, nkeys, seed = 12_500_000, 512, 4096, 5, 2000, 0xFACE0005
    gal = Gallery(gpu_ctx, d, n)
    gal.fill_synthetic(seed=seed, row0=0, n=n, nkeys=nkeys, dup_every=100)
    keys = sg.keys(seed, np.arange(n), nkeys, dup_every=100)
    rng = np.random.default_rng(55)
    dead = rng.integers(0, n, size=20_000)
    gal.tombstone(dead)
    keys[dead] = -1
    # queries: 200 keys in use (about 20 queries each), planted near live rows of their key
    used = rng.choice(nkeys, size=200, replace=False)
    order = np.argsort(keys, kind="stable")
    lo_hi = np.searchsorted(keys[order], np.stack([used, used + 1]))
    rows_of = {int(key): order[lo_hi[0, j]:lo_hi[1, j]] for j, key in enumerate(used)}     # ascending rows
    qkeys = rng.choice(used, size=bq).astype(np.int32)
    targets = np.array([rng.choice(rows_of[int(key)]) for key in qkeys])
    t16, tk = gal.read(targets)
    assert np.array_equal(tk, qkeys)
    q16 = _noisy_queries(rng, t16, 0.75)
    q16[::97] = t16[::97]
    # a tombstoned target must not come back; ANY finds the stored row whatever its key; key 5000 finds nothing
    gone = targets[5]
    gal.tombstone([gone])
    keys[gone] = -1
    rows_of[int(qkeys[5])] = rows_of[int(qkeys[5])][rows_of[int(qkeys[5])] != gone]
    any_idx = np.arange(10, 4096, 512)
    qkeys_sent = qkeys.copy()
    qkeys_sent[any_idx] = -2
    q16[any_idx] = t16[any_idx]
    qkeys_sent[7] = 5000
    got_s, got_r = _search_f16(gal, q16, qkeys_sent, k)
    assert (got_r[7] == -1).all() and np.isneginf(got_s[7]).all()
    assert gone not in got_r[5]
    for i in any_idx:
        twin = targets[i] + 1 if targets[i] % 100 == 98 else (targets[i] - 1 if targets[i] % 100 == 99 else -1)
        live = [t for t in (targets[i], twin) if t >= 0 and keys[t] >= 0]
        assert list(got_r[i, :len(live)]) == sorted(live) and got_s[i, 0] > 0.998
    with np.errstate(invalid="ignore"):
        steps = np.diff(got_s, axis=1)
    assert (np.isnan(steps) | (steps <= 0)).all()             # -inf tails: inf - inf
    qf = q16.astype(np.float32)
    keyed = np.setdiff1d(np.arange(bq), np.concatenate([any_idx, [7]]))
    for key in used:
        qi = keyed[qkeys[keyed] == key]
        if len(qi) == 0:
            continue
        rows = rows_of[int(key)]
        sub16, subk = gal.read(rows)
        assert (subk == key).all()
        ref_s, ref_loc = osr.topk_stored(q16[qi], sub16, k)
        ref_r = np.where(ref_loc >= 0, rows[np.maximum(ref_loc, 0)], -1)
        pos = {int(r): j for j, r in enumerate(rows)}

        def score(i, row, qi=qi, sub16=sub16, pos=pos):
            assert row in pos, f"row {row} does not carry the query's key"
            return float(qf[qi[i]] @ sub16[pos[row]].astype(np.float32))

        _check_lists(got_s[qi], got_r[qi], ref_s, ref_r, score)
    st_s, st_r = _search_f16(gal, q16, qkeys_sent, k, static=True)
    assert np.array_equal(st_r, got_r) and np.array_equal(st_s, got_s)
    gal.close()


def test_config5_rank_shape_collections_stored_together(gpu_ctx):
    """The same rank share (12.5 M x 512, 2,000 keys, 4,096 keyed queries of top-5) with every
    collection's rows stored one after the other, as the reference's per-collection storage keeps
    them (qdrant_database_FE/app.py:108-128): the by-key search reads only each key's tiles and
    answers exactly as the scan of the whole gallery does; a sample also against the oracle over the
    key's rows."""
    import time
    from frhot.engine import Gallery
    from oracle import search as osr
    n, d, bq, k, run, seed = 12_500_000, 512, 4096, 5, 6250, 0xFACE0055
    nkeys = n // run
    gal = Gallery(gpu_ctx, d, n)
    gal.fill_synthetic(seed=seed, row0=0, n=n, nkeys=-run, dup_every=100)
    rng = np.random.default_rng(56)
    dead = rng.integers(0, n, size=20_000)
    gal.tombstone(dead)
    zipf = 1.0 / np.arange(1, nkeys + 1)
    qkeys = rng.choice(nkeys, size=bq, p=zipf / zipf.sum()).astype(np.int32)
    targets = qkeys.astype(np.int64) * run + rng.integers(0, run, size=bq)
    targets = targets[~np.isin(targets, dead)]
    qkeys = (targets // run).astype(np.int32)
    t16, tk = gal.read(targets)
    assert np.array_equal(tk, qkeys)
    q = _noisy_queries(rng, t16, 0.75).astype(np.float32)
    qkeys[3] = 5000                                         # nobody has it
    out, took = {}, {}
    for mode in (Gallery.SEARCH_WHOLE, Gallery.SEARCH_BY_KEY, Gallery.SEARCH_AUTO):
        gal.search_mode(mode)
        gal.search(q[:8], qkeys[:8], k)
        t0 = time.perf_counter()
        out[mode] = gal.search(q, qkeys, k)
        took[mode] = time.perf_counter() - t0
    print(f"config 5 share, {len(qkeys)} keyed queries over {len(set(qkeys.tolist()))} keys: whole gallery "
          f"{took[Gallery.SEARCH_WHOLE] * 1e3:.1f} ms, by key {took[Gallery.SEARCH_BY_KEY] * 1e3:.1f} ms, "
          f"cost rule {took[Gallery.SEARCH_AUTO] * 1e3:.1f} ms")
    for mode in (Gallery.SEARCH_BY_KEY, Gallery.SEARCH_AUTO):
        assert np.array_equal(out[mode][1], out[Gallery.SEARCH_WHOLE][1])
        assert np.array_equal(out[mode][0], out[Gallery.SEARCH_WHOLE][0])
    got_s, got_r = out[Gallery.SEARCH_BY_KEY]
    assert (got_r[3] == -1).all() and np.isneginf(got_s[3]).all()
    live = np.ones(n, bool)
    live[dead] = False
    for i in rng.choice(len(qkeys), size=24, replace=False):
        if i == 3:
            continue
        rows = np.arange(qkeys[i] * run, (qkeys[i] + 1) * run)
        rows = rows[live[rows]]
        sub16, subk = gal.read(rows)
        assert (subk == qkeys[i]).all()
        q16 = osr.unit_rows(q[i:i + 1]).astype(np.float16)
        ref_s, ref_loc = osr.topk_stored(q16, sub16, k)
        ref_r = np.where(ref_loc >= 0, rows[np.maximum(ref_loc, 0)], -1)
        pos = {int(r): j for j, r in enumerate(rows)}
        qf = q16.astype(np.float32)

        def score(_, row, qf=qf, sub16=sub16, pos=pos):
            assert row in pos, f"row {row} does not carry the query's key"
            return float(qf[0] @ sub16[pos[row]].astype(np.float32))

        _check_lists(got_s[i:i + 1], got_r[i:i + 1], ref_s, ref_r, score)
    assert took[Gallery.SEARCH_BY_KEY] < 0.5 * took[Gallery.SEARCH_WHOLE]
    gal.close()
