"""The N > 1 path on CPU: world_size-2 and -3 gloo runs of the row-sharded search orchestration
(all-gather queries -> local scan -> all-gather top-k -> merge) with the scan and merge
played by the CPU oracle; the result must equal one exact search over the whole gallery."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_rows, dim, batch, k, out_dir, keyed=False, always=False):
    for p in (ROOT, os.path.join(ROOT, "face-recognition_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from frhot import synth
        from frhot.dist import ShardedSearch, shard_rows
        from oracle import search as osr
        g = synth.unit_gallery(seed=123, n=n_rows, d=dim)
        g[n_rows - 1] = g[0]                      # a tie that straddles the two shards
        base, rows = shard_rows(n_rows, world, rank)
        shard = g[base:base + rows]
        rng = np.random.default_rng(1000 + rank)
        targets = rng.integers(0, n_rows, size=batch)
        q_local = synth.planted_queries(rng, g[targets], 0.8)
        if rank == 0:
            q_local[0] = g[0]                     # scores exactly 1.0 against rows 0 and n_rows-1

        keys = (np.arange(n_rows) * 7 % 5).astype(np.int32)         # every rank derives the same keys
        qkeys_local = keys[targets].copy()
        if rank == 0 and batch > 2:
            qkeys_local[1] = -2                                       # any collection
            qkeys_local[2] = 77                                       # a collection nobody has

        def search_local(q_all, s_out, r_out, qkeys_all=None):
            assert (qkeys_all is not None) == keyed
            if keyed:
                s, r = osr.cosine_topk(q_all.numpy(), shard, k, keys=keys[base:base + rows], qkeys=qkeys_all.numpy())
            else:
                s, r = osr.cosine_topk(q_all.numpy(), shard, k)
            r = np.where(r >= 0, r + base, -1)
            s_out.copy_(torch.from_numpy(s))
            r_out.copy_(torch.from_numpy(r))

        def merge(s_all, r_all, out_s, out_r):
            # strided views into the gathered blocks: rows first, scores behind them, block after block
            assert s_all.shape[0] == 1 or s_all.stride(0) * 4 == r_all.stride(0) * 8 == ss.block_bytes
            assert s_all.data_ptr() == r_all.data_ptr() + r_all.shape[1] * r_all.shape[2] * 8
            P, n, kk = s_all.shape
            cs = s_all.permute(1, 0, 2).reshape(n, P * kk).numpy()
            cr = r_all.permute(1, 0, 2).reshape(n, P * kk).numpy()
            cs = np.where(cr < 0, -np.inf, cs)
            crs = np.where(cr < 0, np.iinfo(np.int64).max, cr)
            order = np.lexsort((crs, -cs), axis=1)[:, :kk]
            out_s.copy_(torch.from_numpy(np.take_along_axis(cs, order, 1)))
            out_r.copy_(torch.from_numpy(np.take_along_axis(cr, order, 1)))

        ss = ShardedSearch(world, batch, dim, k, torch.device("cpu"), search_local, merge, qdtype=torch.float32,
                           drain_before_exchange=(rank == 0),      # no effect on host tensors: must not disturb the exchange
                           always_exchange=always)
        s, r = ss(torch.from_numpy(q_local), torch.from_numpy(qkeys_local)) if keyed else ss(torch.from_numpy(q_local))
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), s=s.numpy(), r=r.numpy(), q=q_local, targets=targets,
                 qkeys=qkeys_local)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_rows", [(2, 1000), (2, 1001), (3, 1000), (3, 4)])
def test_sharded_search_equals_global_search(tmp_path, world, n_rows):
    """World 3 splits 1000 rows 334 / 334 / 332 (uneven) and 4 rows 2 / 2 / 0 (an empty last
    shard, fewer rows than k: lists padded with -1 on every shard)."""
    dim, batch, k = 64, 6, 5
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_rows, dim, batch, k, str(tmp_path)), nprocs=world, join=True)
    from frhot import synth
    from oracle import search as osr
    g = synth.unit_gallery(seed=123, n=n_rows, d=dim)
    g[n_rows - 1] = g[0]
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    q_all = np.concatenate([o["q"] for o in outs])
    ref_s, ref_r = osr.cosine_topk(q_all, g, k)
    for o in outs:                                  # every rank holds the full answer
        assert np.array_equal(o["r"], ref_r)
        live = ref_r >= 0
        assert np.abs(o["s"][live] - ref_s[live]).max() < 1e-6 and np.isneginf(o["s"][~live]).all()
    assert list(outs[0]["r"][0, :2]) == [0, n_rows - 1]          # tie ordered by ascending global row
    targets = np.concatenate([o["targets"] for o in outs])
    if n_rows >= 1000:
        assert (ref_r[1:, 0] == targets[1:]).mean() > 0.8
    else:
        assert (ref_r[:, n_rows:] == -1).all() and (ref_r[:, :n_rows] >= 0).all()


@pytest.mark.parametrize("world,batch,k", [(3, 5, 5), (3, 1, 5), (3, 3, 5), (3, 5, 3), (1, 1, 5)])
def test_sharded_search_odd_entry_counts(tmp_path, world, batch, k):
    """world * batch * k odd: a rank's block (int64 rows, then float32 scores) no longer ends on an
    8-byte boundary by itself; the exchange pads it.  World 1 runs the exchange as a rehearsal does."""
    n_rows, dim = 1000, 64
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_rows, dim, batch, k, str(tmp_path), False, True), nprocs=world, join=True)
    from frhot import synth
    from oracle import search as osr
    g = synth.unit_gallery(seed=123, n=n_rows, d=dim)
    g[n_rows - 1] = g[0]
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    q_all = np.concatenate([o["q"] for o in outs])
    ref_s, ref_r = osr.cosine_topk(q_all, g, k)
    for o in outs:
        assert np.array_equal(o["r"], ref_r) and np.abs(o["s"] - ref_s).max() < 1e-6


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_search_with_keys_equals_global_search(tmp_path, world):
    """The same with a key per query (the collection it is looked up in): the keys travel with the
    queries (one more small all-gather) and every shard filters by them; ANY and a key nobody has
    included."""
    n_rows, dim, batch, k = 1000, 64, 6, 5
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_rows, dim, batch, k, str(tmp_path), True), nprocs=world, join=True)
    from frhot import synth
    from oracle import search as osr
    g = synth.unit_gallery(seed=123, n=n_rows, d=dim)
    g[n_rows - 1] = g[0]
    keys = (np.arange(n_rows) * 7 % 5).astype(np.int32)
    outs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    q_all = np.concatenate([o["q"] for o in outs])
    qk_all = np.concatenate([o["qkeys"] for o in outs])
    ref_s, ref_r = osr.cosine_topk(q_all, g, k, keys=keys, qkeys=qk_all)
    for o in outs:
        assert np.array_equal(o["r"], ref_r)
        live = ref_r >= 0
        assert np.abs(o["s"][live] - ref_s[live]).max() < 1e-6 and np.isneginf(o["s"][~live]).all()
    assert (ref_r[2] == -1).all()
    rows = ref_r[ref_r >= 0]
    assert (keys[ref_r[3:][ref_r[3:] >= 0]] == np.repeat(qk_all[3:], k)[(ref_r[3:] >= 0).ravel()]).all() and rows.size


def test_shard_rows_cover_everything():
    from frhot.dist import shard_rows
    for total in (1, 7, 8, 1000, 10_000_000):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_rows(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(n for _, n in spans) == total
            for (b0, n0), (b1, _n1) in zip(spans, spans[1:]):
                assert b1 == b0 + n0 or _n1 == 0
