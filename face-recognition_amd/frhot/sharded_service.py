"""The DB service's operations (qdrant_database_FE/app.py:108-128,242-297,300-386,409-438) over a
gallery that is ROW-SHARDED across the GPUs of one node, one process per GPU.

The reference scales by replicas: every uvicorn worker talks to the one Qdrant server
(docker-compose.yml:34,53).  Here the points themselves are spread over P galleries, so P GPUs
hold P times the rows and every search reads all of them at once:

  * placement: a point's global row is shard * capacity_per_shard + local row.  The rows of one
    key (collection + store) are STRIPED over the shards, point by point in insertion order
    (the n-th point of a key goes to shard n mod P, the next shard with room if that one is
    full); inside its shard a key still takes whole 256-row tiles.  A search inside a collection
    therefore costs every shard 1/P of that collection's tiles, and a large collection does not
    fill one GPU while the others stay empty.
  * the host tables (collections, row -> point, rows per payload id, row allocators of ALL
    shards) are REPLICATED: every rank applies every mutation, in the same order, so the ten-
    faces-per-id eviction (app.py:207-240) and delete-by-id (app.py:409-438) decide the same on
    every rank; only the owner of a row touches device memory.  `mutate` is the collective that
    brings every rank's pending mutations into one order (rank order, then list order).
  * search: every rank embeds / normalises ITS requests (data parallel, no collective), then the
    exchange of frhot.dist.ShardedSearch runs (queries and keys gathered, every shard searched
    by key with its row base added, answers gathered and merged), and every rank votes on its
    own requests' hits from the replicated table (app.py:340-384, unchanged).

Every method marked COLLECTIVE must be called by all ranks, in the same order (with one rank and
no process group they are plain calls).  No scaling curve has been measured for this path: it
is covered by gloo runs on CPU and by rehearsals of several ranks on one GPU.
"""
from __future__ import annotations

import datetime
import os
import uuid

import numpy as np
import torch
import torch.distributed as dist

from .dist import ShardedSearch
from .gallery_service import SEARCH_LIMIT, GalleryService, _RowAllocator
from .vote import vote


def unit_f16(vecs: np.ndarray) -> np.ndarray:
    """Rows normalised as the library normalises them on upload (double-precision sum of squares,
    norm rounded to float32, one float32 division per element, rounded to fp16)."""
    x = np.asarray(vecs, dtype=np.float32)
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=-1, keepdims=True)).astype(np.float32)
    n = np.where(n < np.float32(1e-30), np.float32(1.0), n)
    return (x / n).astype(np.float16)


class HipShardBackend:
    """One rank's GPU side: a frhot Context and its Gallery of capacity_per_shard rows.
    shared_stream: the context was created on the torch stream that is current whenever the service
    runs (Context(stream=torch_stream.cuda_stream)), so library calls, torch copies and the
    collectives are ordered by that one stream; otherwise every hand-over between torch and the
    library waits on the host."""

    def __init__(self, ctx, gallery, rank: int, capacity_per_shard: int, device, shared_stream: bool = False):
        self.ctx, self.gallery, self.rank, self.cap, self.device = ctx, gallery, rank, capacity_per_shard, device
        self.qdtype = torch.float16
        self.shared = shared_stream

    def _to_lib(self):          # torch's work must have landed before the library reads it
        if not self.shared:
            torch.cuda.current_stream(self.device).synchronize()

    def _to_torch(self):        # and the library's before torch (or a collective) reads it
        if not self.shared:
            self.ctx.sync()

    def unit_queries(self, vecs: np.ndarray) -> torch.Tensor:
        return torch.from_numpy(unit_f16(vecs)).to(self.device)

    def embed(self, crops, gamma: bool) -> torch.Tensor:
        from . import synth
        from .engine import Context
        buf, offsets, hw, strides = synth.pack_crops(crops)
        crops_dev = torch.from_numpy(buf).to(self.device)
        q = torch.empty((len(crops), self.gallery.dim), dtype=torch.float16, device=self.device)
        self._to_lib()
        self.ctx.embed_dev(crops_dev.data_ptr(), offsets, hw, strides, Context.flags(gamma=gamma, flip=True), None,
                           q.data_ptr(), self.gallery.dim)
        self._to_torch()
        self._keep = crops_dev          # until the stream has passed the call
        return q

    def search_local(self, q_all, s_out, r_out, qkeys_all=None):
        keys = None if qkeys_all is None else qkeys_all.cpu().numpy()
        self._to_lib()
        self.gallery.search_dev_keyed(q_all.data_ptr(), keys, q_all.shape[0], s_out.shape[1], s_out.data_ptr(),
                                      r_out.data_ptr(), self.rank * self.cap)
        self._to_torch()

    def merge(self, s_all, r_all, out_s, out_r, block_bytes):
        self._to_lib()
        self.ctx.merge_topk_packed_dev(r_all.data_ptr(), block_bytes, r_all.shape[0], r_all.shape[1], r_all.shape[2],
                                       out_s.data_ptr(), out_r.data_ptr())
        self._to_torch()


class _ShardedRows:
    """Row allocators of all shards (replicated on every rank): stripes a key's points over the
    shards and hands out global rows."""

    def __init__(self, world: int, capacity_per_shard: int):
        self.P, self.cap = world, capacity_per_shard
        self.capacity = world * capacity_per_shard
        self._shards = [_RowAllocator(capacity_per_shard) for _ in range(world)]
        self._count: dict[int, int] = {}            # key -> points ever placed

    def free_count(self) -> int:
        return sum(a.free_count() for a in self._shards)

    def alloc(self, key: int) -> int:
        n = self._count.get(key, 0)
        for step in range(self.P):
            sh = (n + step) % self.P
            try:
                local = self._shards[sh].alloc(key)
            except MemoryError:
                continue
            self._count[key] = n + 1
            return sh * self.cap + local
        raise MemoryError("gallery capacity exhausted")

    def release(self, row: int):
        self._shards[row // self.cap].release(row % self.cap)


class _ShardedGallery:
    """What GalleryService sees as its gallery: global rows in, the owner's device memory touched."""

    def __init__(self, backend, rank: int, world: int, capacity_per_shard: int, dim: int, batch: int,
                 host_staged: bool = False):
        self.b, self.rank, self.P, self.cap, self.dim, self.batch = backend, rank, world, capacity_per_shard, dim, batch
        self.exchange = ShardedSearch(
            world, batch, dim, SEARCH_LIMIT, backend.device, backend.search_local,
            lambda s_all, r_all, out_s, out_r: backend.merge(s_all, r_all, out_s, out_r, self.exchange.block_bytes),
            qdtype=backend.qdtype, host_staged=host_staged)

    def _mine(self, rows):
        rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        return rows, (rows // self.cap) == self.rank

    def upsert(self, vecs, keys, rows):
        rows, mine = self._mine(rows)
        if mine.any():
            vecs = np.asarray(vecs, np.float32).reshape(len(rows), self.dim)
            keys = np.broadcast_to(np.asarray(keys, np.int32), rows.shape)
            self.b.gallery.upsert(vecs[mine], keys[mine], rows[mine] % self.cap)

    def write_f16(self, rows_f16, keys, rows):
        rows, mine = self._mine(rows)
        if mine.any():
            keys = np.broadcast_to(np.asarray(keys, np.int32), rows.shape)
            self.b.gallery.write_f16(np.asarray(rows_f16, np.float16)[mine], keys[mine], rows[mine] % self.cap)

    def tombstone(self, rows):
        rows, mine = self._mine(rows)
        if mine.any():
            self.b.gallery.tombstone(rows[mine] % self.cap)

    def read(self, rows):
        """COLLECTIVE: the stored fp16 rows and keys of global rows, from whichever shard holds them."""
        rows, mine = self._mine(rows)
        vec = np.zeros((len(rows), self.dim), np.float16)
        keys = np.full(len(rows), -1, np.int32)
        part = self.b.gallery.read(rows[mine] % self.cap) if mine.any() else None
        parts = [part]
        if self.P > 1:
            parts = [None] * self.P
            dist.all_gather_object(parts, part)
        for r, pt in enumerate(parts):
            if pt is not None:
                sel = (rows // self.cap) == r
                vec[sel], keys[sel] = pt[0], pt[1]
        return vec, keys

    def search_queries(self, q_local: torch.Tensor, keys: np.ndarray):
        """COLLECTIVE: this rank's unit queries [n <= batch, D] (device) and their keys -> (scores,
        rows) [n, k] of global rows, on the host."""
        n = q_local.shape[0]
        if n > self.batch:
            raise ValueError(f"{n} queries in one step; the service was built for {self.batch}")
        q = torch.zeros((self.batch, self.dim), dtype=q_local.dtype, device=q_local.device)
        q[:n] = q_local
        k = np.full(self.batch, GalleryService.NO_KEY, np.int32)       # padding matches nothing
        k[:n] = keys
        s, r = self.exchange(q, torch.from_numpy(k).to(q_local.device))
        lo = self.rank * self.batch if self.exchange.exchange else 0
        return s[lo:lo + n].cpu().numpy(), r[lo:lo + n].cpu().numpy()

    def close(self):
        self.b.gallery.close()


class ShardedGalleryService(GalleryService):
    _always_search = True
    _search_holds_lock = True       # a search is a collective: the ranks' calls must come in one order

    def __init__(self, backend, rank: int, world: int, capacity_per_shard: int, dim: int, batch: int = 64,
                 host_staged: bool = False):
        """backend: HipShardBackend (or a stand-in with its methods); batch: the most requests one
        rank brings to one search step (every step exchanges world * batch queries);
        host_staged: see frhot.dist.ShardedSearch (a gloo process group over device buffers)."""
        if world > 1 and not dist.is_initialized():
            raise RuntimeError("ShardedGalleryService with more than one rank needs an initialised process group")
        self.rank, self.world, self.batch = rank, world, batch
        self.backend = backend
        super().__init__(None, capacity=world * capacity_per_shard, dim=dim,
                         gallery=_ShardedGallery(backend, rank, world, capacity_per_shard, dim, batch,
                                                 host_staged))
        self._rows = _ShardedRows(world, capacity_per_shard)

    # ------------------------------------------------------------- mutations
    def mutate(self, ops):
        """COLLECTIVE.  ops: this rank's pending mutations, each ("create_collection", name),
        ("insert_point", {keyword arguments of insert_point}) or ("delete_point", collection, id).
        Every rank applies all ranks' mutations, rank by rank; returns the answers to this rank's."""
        mine = []
        for op in ops:
            if op[0] == "insert_point":
                kw = dict(op[1])
                kw.setdefault("time_created", None)
                if kw["time_created"] is None:
                    kw["time_created"] = datetime.datetime.now().strftime("%Y/%m/%d %H:%M:%S")
                kw.setdefault("uid", str(uuid.uuid4()))
                op = ("insert_point", kw)
            mine.append(op)
        every = [mine]
        if self.world > 1:
            every = [None] * self.world
            dist.all_gather_object(every, mine)
        answers = []
        for r, lst in enumerate(every):
            for op in lst:
                if op[0] == "create_collection":
                    res = self.create_collection(op[1])
                elif op[0] == "insert_point":
                    res = GalleryService.insert_point(self, **op[1])
                elif op[0] == "delete_point":
                    res = GalleryService.delete_point(self, op[1], op[2])
                else:
                    res = (404, {"message": f"unknown mutation {op[0]!r}"})
                if r == self.rank:
                    answers.append(res)
        return answers

    def insert_point(self, *args, **kwargs):
        """COLLECTIVE form of one insert: every rank makes this same call (one rank's own inserts go
        through mutate)."""
        if self.world > 1 and kwargs.get("uid") is None and not kwargs.get("is_update_id"):
            raise RuntimeError("insert_point on a sharded service: use mutate(), which agrees on the point's id")
        return GalleryService.insert_point(self, *args, **kwargs)

    # ---------------------------------------------------------------- search
    def _search_batch(self, vecs, keys):
        q = (self.backend.unit_queries(np.stack(vecs)) if vecs
             else torch.zeros((0, self.dim), dtype=self.backend.qdtype, device=self.backend.device))
        return self.gallery.search_queries(q, np.asarray(keys, np.int32))

    # search_point / search_points: COLLECTIVE (inherited; every rank calls, with its own requests or none)

    def recognise_points(self, requests, want_embeddings: bool = False, gamma: bool = True,
                         want_sharpness: bool = False):
        """COLLECTIVE.  This rank's [(collection_name, crop, store_id)] -> [(status, content, None)]:
        crops embedded on this rank's GPU, descriptors exchanged on the devices, every shard searched."""
        if want_embeddings or want_sharpness:
            raise NotImplementedError("the sharded step returns answers only")
        with self._lock:
            out = [None] * len(requests)
            keys, bad = [], {}
            for i, (col, _crop, store_id) in enumerate(requests):
                if col is None or col == "" or col not in self._collections:
                    bad[i] = (404, {"message": "Collection name not found or invalid!"})
                k = self._keys.get((col, store_id))
                keys.append(self.NO_KEY if (k is None or i in bad) else k)
            q = (self.backend.embed([r[1] for r in requests], gamma) if requests
                 else torch.zeros((0, self.dim), dtype=self.backend.qdtype, device=self.backend.device))
            scores, rows = self.gallery.search_queries(q, np.asarray(keys, np.int32))
            for i in range(len(requests)):
                if i in bad:
                    out[i] = bad[i] + (None,)
                elif keys[i] == self.NO_KEY:
                    out[i] = (200, {"message": "Point not found", "data": []}, None)
                else:
                    out[i] = (200, vote(self._hits(scores[i], rows[i])), None)
            return out

    # -------------------------------------------------------------- snapshots
    def create_snapshot(self, collection_name, directory):
        """COLLECTIVE (app.py:130-141).  Rank 0 writes the file, in the single-gallery format: a
        snapshot taken from P shards restores into one gallery and the other way round."""
        if self.rank == 0:
            return GalleryService.create_snapshot(self, collection_name, directory)
        with self._lock:                            # the same collective read, nothing written
            if collection_name not in self._collections:
                return 404, {"message": "Collection name not found or invalid!"}
            rows = sorted(r for r, p in self._points.items() if p["collection"] == collection_name)
            if rows:
                self.gallery.read(np.asarray(rows, dtype=np.int64))
            self._seq += 1
            return 200, {"name": None, "size": None, "points": len(rows)}

    # recover_snapshot: COLLECTIVE (inherited): every rank reads the same file and applies it.
