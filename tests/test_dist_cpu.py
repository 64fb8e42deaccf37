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


# ---------------------------------------------------------------- sharded service
def _service_trace(seed, dim, ticks, world):
    """A random enrol / evict / delete / search trace, tick by tick: per tick and rank a list of
    mutations and a list of search requests.  Every enrolled vector is distinct (a person's
    direction plus noise), so no two rows tie and the answer does not depend on row numbers."""
    rng = np.random.default_rng(seed)
    people = {}                                     # (collection, id) -> unit direction

    def person(col, pid):
        if (col, pid) not in people:
            v = rng.standard_normal(dim).astype(np.float32)
            people[(col, pid)] = v / np.linalg.norm(v)
        return people[(col, pid)]

    cols = ["S1_Employees", "S1_Customers", "S2_Employees"]
    trace, clock = [], 0
    for t in range(ticks):
        tick = []
        for r in range(world):
            muts, reqs = [], []
            if t == 0 and r == 0:
                muts += [("create_collection", c) for c in cols] + [("create_collection", cols[0])]
            for _ in range(int(rng.integers(0, 4))):
                col = cols[int(rng.integers(0, len(cols)))]
                pid = str(int(rng.integers(0, 3)))              # few ids: the ten-faces limit is reached
                kind = rng.random()
                if kind < 0.8 and t > 0:
                    clock += 1
                    vec = person(col, pid) + 0.15 * rng.standard_normal(dim).astype(np.float32) / np.sqrt(dim)
                    muts.append(("insert_point", dict(collection_name=col, vector_embedding=vec.tolist(), id=pid,
                                                      name=f"name{pid}", store_id=col.split("_")[0],
                                                      time_created=f"2026/01/01 00:{clock // 60:02d}:{clock % 60:02d}")))
                elif kind < 0.9 and t > 0:
                    muts.append(("delete_point", col, pid))
                elif t > 0:
                    muts.append(("insert_point", dict(collection_name="nowhere", vector_embedding=[0.0] * dim, id=pid,
                                                      name="x", store_id="S1")))      # 404: unknown collection
            for _ in range(int(rng.integers(0, 4))):
                col = cols[int(rng.integers(0, len(cols)))] if rng.random() < 0.9 else "S9_Employees"
                pid = str(int(rng.integers(0, 4)))
                q = person(col, pid) + 0.3 * rng.standard_normal(dim).astype(np.float32) / np.sqrt(dim)
                reqs.append((col, q.tolist(), col.split("_")[0] if rng.random() < 0.9 else "S7"))
            tick.append((muts, reqs))
        trace.append(tick)
    return trace


class _NumpyShardBackend:
    """CPU stand-in for frhot.sharded_service.HipShardBackend (the oracle's search on a numpy shard)."""

    def __init__(self, rank, cap, dim):
        import torch as _t
        from fakes import NumpyGallery
        self.rank, self.cap = rank, cap
        self.gallery = NumpyGallery(dim, cap)
        self.device, self.qdtype = _t.device("cpu"), _t.float32

    def unit_queries(self, vecs):
        from oracle import search as osr
        return torch.from_numpy(osr.unit_rows(vecs))

    def search_local(self, q_all, s_out, r_out, qkeys_all=None):
        s, r = self.gallery.search(q_all.numpy(), qkeys_all.numpy(), s_out.shape[1])
        s_out.copy_(torch.from_numpy(s))
        r_out.copy_(torch.from_numpy(np.where(r >= 0, r + self.rank * self.cap, -1)))

    def merge(self, s_all, r_all, out_s, out_r, block_bytes):
        P, n, kk = s_all.shape
        cs = s_all.permute(1, 0, 2).reshape(n, P * kk).numpy()
        cr = r_all.permute(1, 0, 2).reshape(n, P * kk).numpy()
        cs = np.where(cr < 0, -np.inf, cs)
        order = np.lexsort((np.where(cr < 0, np.iinfo(np.int64).max, cr), -cs), axis=1)[:, :kk]
        out_s.copy_(torch.from_numpy(np.take_along_axis(cs, order, 1)))
        out_r.copy_(torch.from_numpy(np.take_along_axis(cr, order, 1)))


def _service_worker(rank, world, port, seed, dim, ticks, cap, out_dir):
    for p in (ROOT, os.path.join(ROOT, "face-recognition_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import json
        from frhot.sharded_service import ShardedGalleryService
        svc = ShardedGalleryService(_NumpyShardBackend(rank, cap, dim), rank, world, cap, dim, batch=4)
        answers = []
        for tick in _service_trace(seed, dim, ticks, world):
            muts, reqs = tick[rank]
            answers.append({"mut": svc.mutate(muts), "search": svc.search_points(reqs)})
        snap = svc.create_snapshot("S1_Employees", os.path.join(out_dir, "snaps"))
        counts = {c: svc.point_count(c) for c in svc.get_collections()["collections"]}
        shards = sorted({r // cap for r in svc._points})
        with open(os.path.join(out_dir, f"svc{rank}.json"), "w") as f:
            json.dump({"answers": answers, "counts": counts, "shards": shards, "snap": snap}, f)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_service_answers_like_the_single_service(tmp_path, world):
    """The same trace of enrolments (with the ten-faces-per-id eviction), deletions and searches
    through a ShardedGalleryService of 2 or 3 ranks (gloo, numpy shards) and through ONE
    GalleryService: every mutation and every search is answered identically, the points of a
    collection end up striped over all shards, and the snapshot rank 0 writes restores into a
    single gallery that then answers the same."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "face-recognition_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fakes import NumpyGallery
    from frhot.gallery_service import GalleryService
    seed, dim, ticks, cap = 77, 32, 90, 600
    port = _free_port()
    mp.spawn(_service_worker, args=(world, port, seed, dim, ticks, cap, str(tmp_path)), nprocs=world, join=True)
    outs = [json.load(open(tmp_path / f"svc{r}.json")) for r in range(world)]
    single = GalleryService(None, capacity=world * cap, dim=dim, gallery=NumpyGallery(dim, world * cap))
    n_found = n_evicting = 0
    for t, tick in enumerate(_service_trace(seed, dim, ticks, world)):
        # mutations of a tick apply in rank order, then every rank searches
        for r, (muts, _reqs) in enumerate(tick):
            want = []
            for op in muts:
                if op[0] == "create_collection":
                    want.append(single.create_collection(op[1]))
                elif op[0] == "insert_point":
                    n_evicting += len(single._by_id.get((op[1]["collection_name"], op[1]["id"]), [])) >= 10
                    want.append(single.insert_point(**op[1]))
                else:
                    want.append(single.delete_point(op[1], op[2]))
            assert json.loads(json.dumps(want)) == outs[r]["answers"][t]["mut"]
        for r, (_muts, reqs) in enumerate(tick):
            want = json.loads(json.dumps(single.search_points(reqs)))
            got = outs[r]["answers"][t]["search"]
            assert len(want) == len(got)
            for w, g in zip(want, got):
                assert w[0] == g[0] and w[1]["message"] == g[1]["message"]
                if w[1].get("data"):
                    n_found += 1
                    assert w[1]["data"][0][1] == g[1]["data"][0][1]                    # the same payload
                    assert abs(w[1]["data"][0][0] - g[1]["data"][0][0]) < 1e-6
    assert n_found > 20 and n_evicting > 3
    counts = {c: single.point_count(c) for c in single.get_collections()["collections"]}
    for o in outs:
        assert o["counts"] == counts and o["shards"] == list(range(world))
    # rank 0's snapshot, restored into a fresh single gallery
    snap = outs[0]["snap"]
    assert snap[0] == 200 and snap[1]["points"] == counts["S1_Employees"]
    fresh = GalleryService(None, capacity=world * cap, dim=dim, gallery=NumpyGallery(dim, world * cap))
    st, _ = fresh.recover_snapshot("S1_Employees", str(tmp_path / "snaps" / "S1_Employees" / snap[1]["name"]))
    assert st == 200 and fresh.point_count("S1_Employees") == counts["S1_Employees"]
    q = [(c, v, s) for tick in _service_trace(seed, dim, ticks, world) for _m, reqs in tick for (c, v, s) in reqs
         if c == "S1_Employees"][:20]
    for a, b in zip(single.search_points(q), fresh.search_points(q)):
        assert a[0] == b[0] and a[1]["message"] == b[1]["message"]
        if a[1].get("data"):
            assert a[1]["data"][0][1] == b[1]["data"][0][1]
