"""The search entry point that allocates nothing (frhot_search_dev_static), through the C ABI,
against frhot_search_dev on the same inputs: bit-identical scores and rows, on two calls in a
row, the second with a smaller input (the static scratch keeps the first call's contents)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(gal, q16, qk, bq, k, base):
    import torch
    dev = q16.device
    out = []
    for fn in (gal.search_dev, gal.search_dev_static):
        s = torch.full((bq, k), 7.0, dtype=torch.float32, device=dev)
        r = torch.full((bq, k), 7, dtype=torch.int64, device=dev)
        fn(q16.data_ptr(), None if qk is None else qk.data_ptr(), bq, k, s.data_ptr(), r.data_ptr(), base)
        gal.ctx.sync()
        out.append((s.cpu().numpy(), r.cpu().numpy()))
    return out


def test_static_scratch_search_equals_search_dev(gpu_ctx):
    import torch
    from frhot import FrhotError, synth
    from frhot.engine import Gallery
    from oracle import search as osr
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(123)
    n, d = 700_000, 64
    g = rng.standard_normal((n, d), dtype=np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    g[500_001] = g[17]                                         # a tie far apart in the scan order
    keys = rng.integers(0, 3, size=n).astype(np.int32)
    gal = Gallery(gpu_ctx, d, n)
    for lo in range(0, n, 100_000):
        gal.upsert(g[lo:lo + 100_000], keys[lo:lo + 100_000], np.arange(lo, min(n, lo + 100_000)))
    dead = rng.integers(0, n, size=3000)
    gal.tombstone(dead)
    keys_live = keys.copy()
    keys_live[dead] = -1
    # first call: 600 queries (three 256-query tiles, score bounds on), top-5, keyed
    bq1 = 600
    t1 = rng.integers(0, n, size=bq1)
    q1 = synth.planted_queries(rng, g[t1], 0.7)
    q1[0] = g[17]
    q16 = torch.from_numpy(osr.unit_rows(q1).astype(np.float16)).to(dev)
    qk1 = keys[t1].copy()
    qk1[::5] = -2
    qk = torch.from_numpy(qk1).to(dev)
    (s_a, r_a), (s_b, r_b) = _both(gal, q16, qk, bq1, 5, 1000)
    assert np.array_equal(r_a, r_b) and np.array_equal(s_a.view(np.uint32), s_b.view(np.uint32))
    assert (r_a[:, 0] >= 1000).all()
    # second call, smaller: 40 queries, top-10, no keys (another kernel instantiation, fewer lists)
    bq2 = 40
    q16b = q16[100:100 + bq2].contiguous()
    (s_a, r_a), (s_b, r_b) = _both(gal, q16b, None, bq2, 10, 0)
    assert np.array_equal(r_a, r_b) and np.array_equal(s_a.view(np.uint32), s_b.view(np.uint32))
    ref_s, ref_r = osr.cosine_topk(q1[100:100 + bq2], g, 10, keys=keys_live, qkeys=np.full(bq2, -2), gallery_dtype=np.float16)
    assert (r_b >= 0).all() and (r_b == ref_r).mean() >= 0.99 and np.abs(s_b - ref_s).max() <= 2e-5
    # third: a single query of top-16 on a tiny gallery, and the empty gallery
    small = Gallery(gpu_ctx, d, 300)
    (s_a, r_a), (s_b, r_b) = _both(small, q16b, None, 3, 16, 0)
    assert (r_b == -1).all() and np.isneginf(s_b).all() and np.array_equal(r_a, r_b)
    small.upsert(g[:300], 0, np.arange(300))
    (s_a, r_a), (s_b, r_b) = _both(small, q16b, None, 1, 16, 5)
    assert np.array_equal(r_a, r_b) and np.array_equal(s_a.view(np.uint32), s_b.view(np.uint32))
    small.close()
    # beyond the static scratch: refused with a clear error, nothing launched
    big = torch.zeros((8200, d), dtype=torch.float16, device=dev)
    s = torch.empty((8200, 5), dtype=torch.float32, device=dev)
    r = torch.empty((8200, 5), dtype=torch.int64, device=dev)
    with pytest.raises(FrhotError) as e:
        gal.search_dev_static(big.data_ptr(), None, 8200, 5, s.data_ptr(), r.data_ptr(), 0)
    assert e.value.code == -4 and "static scratch" in e.value.message
    with pytest.raises(FrhotError):
        gal.search_dev_static(big.data_ptr(), None, 4100, 10, s.data_ptr(), r.data_ptr(), 0)
    gal.close()


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0]


def test_static_search_allocates_no_device_memory(gpu_ctx):
    """The property the entry point exists for: free device memory does not move across
    frhot_search_dev_static calls, each larger than any call before it on a gallery that never ran
    frhot_search_dev, while the same sequence through frhot_search_dev does take memory (its
    candidate lists and bounds grow on demand)."""
    import torch
    from frhot.engine import Gallery
    dev = torch.device("cuda", 0)
    n, d = 300_000, 64
    gal = Gallery(gpu_ctx, d, n)
    gal.fill_synthetic(seed=5, row0=0, n=n, nkeys=0, dup_every=0)
    sizes = [(300, 5), (1000, 10), (2048, 16), (8192, 5)]
    bqmax = max(b for b, _ in sizes)
    q = torch.randn((bqmax, d), device=dev).to(torch.float16)
    s = torch.empty((bqmax, 16), dtype=torch.float32, device=dev)
    r = torch.empty((bqmax, 16), dtype=torch.int64, device=dev)
    # every kernel instantiation once, at small sizes: whatever the runtime sets up lazily at a
    # kernel's first launch is not this entry point's doing
    for bq, k in [(1, 5), (1, 10), (1, 16), (100, 5), (100, 10), (200, 5)]:
        gal.search_dev_static(q.data_ptr(), None, bq, k, s.data_ptr(), r.data_ptr(), 0)
    gpu_ctx.sync()
    before = _free_bytes()
    for bq, k in sizes:
        gal.search_dev_static(q.data_ptr(), None, bq, k, s.data_ptr(), r.data_ptr(), 0)
        gpu_ctx.sync()
        assert _free_bytes() == before, f"static search of {bq} queries, top-{k}, changed free device memory"
    want_r = r.view(-1)[:8192 * 5].clone()          # the last static answer (8192 queries of top-5)
    for bq, k in sizes:
        gal.search_dev(q.data_ptr(), None, bq, k, s.data_ptr(), r.data_ptr(), 0)
    gpu_ctx.sync()
    after = _free_bytes()
    assert after < before - 50 * 8192 * 5, "frhot_search_dev was expected to allocate its candidate lists"
    assert torch.equal(want_r, r.view(-1)[:8192 * 5])
    gal.close()


def test_static_scratch_is_refused_to_a_second_context_while_busy(gpu_ctx):
    """One array per GPU: a second context's call while the first context's searches are still
    running gets FRHOT_ESTATE instead of the array; once the first has finished it gets it."""
    import torch
    from frhot import FrhotError
    from frhot.engine import Context, Gallery
    dev = torch.device("cuda", 0)
    n, d, bq = 2_000_000, 64, 512
    gal = Gallery(gpu_ctx, d, n)
    gal.fill_synthetic(seed=6, row0=0, n=n, nkeys=0, dup_every=0)
    other = Context(device=0, max_batch=1)
    small = Gallery(other, d, 1000)
    small.fill_synthetic(seed=7, row0=0, n=1000, nkeys=0, dup_every=0)
    q = torch.randn((bq, d), device=dev).to(torch.float16)
    s = torch.empty((bq, 5), dtype=torch.float32, device=dev)
    r = torch.empty((bq, 5), dtype=torch.int64, device=dev)
    s2 = torch.empty((4, 5), dtype=torch.float32, device=dev)
    r2 = torch.empty((4, 5), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    try:
        for _ in range(200):                 # about 0.2 ms each: tens of milliseconds of queued work
            gal.search_dev_static(q.data_ptr(), None, bq, 5, s.data_ptr(), r.data_ptr(), 0)
        with pytest.raises(FrhotError) as e:
            small.search_dev_static(q.data_ptr(), None, 4, 5, s2.data_ptr(), r2.data_ptr(), 0)
        assert e.value.code == -3 and "in use" in e.value.message
        gpu_ctx.sync()
        small.search_dev_static(q.data_ptr(), None, 4, 5, s2.data_ptr(), r2.data_ptr(), 0)   # now it is free
        other.sync()
        want_s = torch.empty_like(s2)
        want_r = torch.empty_like(r2)
        small.search_dev(q.data_ptr(), None, 4, 5, want_s.data_ptr(), want_r.data_ptr(), 0)
        other.sync()
        assert torch.equal(r2, want_r) and torch.equal(s2, want_s)
        # and back: the first context takes it over again once the second is idle
        gal.search_dev_static(q.data_ptr(), None, bq, 5, s.data_ptr(), r.data_ptr(), 0)
        gpu_ctx.sync()
    finally:
        gpu_ctx.sync()
        small.close()
        other.close()
        gal.close()
