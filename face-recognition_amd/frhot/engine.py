"""Python handles over the C ABI: Context (model + embed pipeline) and Gallery.

Thin by design: argument marshalling and error translation only.  All
arithmetic happens in libfrhot.so on the GPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import EMBED_DIM, FLAG_GAMMA, FLAG_NOFLIP, KEY_ANY, check


def _ptr(a: np.ndarray, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def _crop_args(images):
    """(objects to keep alive, pointer array, [n,2] heights and widths, row strides) of n crops.
    Addresses come from the arrays' interface dictionaries: one ctypes pointer object per crop
    cost a batch of 256 about a millisecond."""
    arrs = []
    for im in images:
        if type(im) is not np.ndarray:
            im = np.asarray(im)
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise ValueError("crops must be uint8 arrays of shape [h, w, 3] (BGR)")
        st = im.strides
        # (the row stride of a one-row image means nothing: numpy keeps whatever the view had)
        if st[2] != 1 or st[1] != 3 or (im.shape[0] > 1 and st[0] < 3 * im.shape[1]):
            im = np.ascontiguousarray(im)
        arrs.append(im)
    n = len(arrs)
    addr = np.fromiter((a.__array_interface__["data"][0] for a in arrs), dtype=np.uint64, count=n)
    hw = np.array([a.shape[:2] for a in arrs], dtype=np.int32).reshape(n, 2)
    strides = np.fromiter((a.strides[0] if a.shape[0] > 1 else 3 * a.shape[1] for a in arrs), dtype=np.int32, count=n)
    ptrs = addr.ctypes.data_as(C.POINTER(C.POINTER(C.c_uint8)))
    return (arrs, addr), ptrs, hw, strides


def laplacian_variance(sums, hw) -> np.ndarray:
    """Variance of an integer image from its exact sum and sum of squares: (n*S2 - S1^2) / n^2,
    evaluated in whole numbers and rounded once (numpy's .var() of the same values agrees to a
    few units in the last place).  sums: int64 [m, 2]; hw: [m, 2] rows, columns."""
    out = np.empty(len(sums), dtype=np.float64)
    for i, ((s1, s2), (h, w)) in enumerate(zip(np.asarray(sums).tolist(), np.asarray(hw).tolist())):
        n = int(h) * int(w)
        out[i] = (n * s2 - s1 * s1) / (n * n)
    return out


class Context:
    """One GPU, one stream, one loaded model (frhot_ctx)."""

    def __init__(self, device: int = 0, max_batch: int = 256, stream: int | None = None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.frhot_ctx_create(device, C.c_void_p(stream or 0), max_batch, C.byref(self._h)))
        self.device = device
        self.max_batch = max_batch
        self._blob = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.frhot_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def sync(self):
        check(self._lib.frhot_ctx_sync(self._h))

    def load_model(self, blob: np.ndarray, input_scale: float = 1.0 / 16.0):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        check(self._lib.frhot_model_load(self._h, C.c_void_p(blob.ctypes.data), blob.size, input_scale))

    @staticmethod
    def flags(gamma: bool = False, flip: bool = True) -> int:
        return (FLAG_GAMMA if gamma else 0) | (0 if flip else FLAG_NOFLIP)

    def embed(self, images, gamma: bool = False, flip: bool = True) -> np.ndarray:
        """uint8 BGR crops -> float32 [n, 4096] unit-norm descriptors (host in, host out)."""
        n = len(images)
        out = np.empty((n, EMBED_DIM), dtype=np.float32)
        for lo in range(0, n, self.max_batch):
            chunk = images[lo:lo + self.max_batch]
            arrs, ptrs, hw, strides = _crop_args(chunk)
            sub = out[lo:lo + len(chunk)]
            check(self._lib.frhot_embed(self._h, ptrs, _ptr(hw, C.c_int32), _ptr(strides, C.c_int32), len(chunk),
                                        self.flags(gamma, flip), _ptr(sub, C.c_float)))
        return out

    def embed_dev(self, crops_dev: int, offsets: np.ndarray, hw: np.ndarray, strides: np.ndarray, flags: int,
                  out_f32_dev: int | None, out_q_f16_dev: int | None, dq: int = EMBED_DIM):
        """Device-pointer form; enqueues only."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        hw = np.ascontiguousarray(hw, dtype=np.int32)
        strides = np.ascontiguousarray(strides, dtype=np.int32)
        check(self._lib.frhot_embed_dev(self._h, C.c_void_p(crops_dev), _ptr(offsets, C.c_int64),
                                        _ptr(hw, C.c_int32), _ptr(strides, C.c_int32), len(offsets), flags,
                                        C.c_void_p(out_f32_dev or 0), C.c_void_p(out_q_f16_dev or 0), dq))

    def debug_read(self, stage: int, n: int) -> np.ndarray:
        if stage == 0:
            out = np.empty((n, 224, 224, 3), dtype=np.uint8)
        elif stage == 1:
            out = np.empty((n, 226, 226, 4), dtype=np.float16)
        elif 10 <= stage < 23:
            from .weights import LAYERS
            pooled = {1, 3, 6, 9, 12}
            hw = [224, 224, 112, 112, 56, 56, 56, 28, 28, 28, 14, 14, 14][stage - 10]
            ho = hw // 2 if (stage - 10) in pooled else hw
            p = ho + (0 if stage == 22 else 2)
            out = np.empty((n, p, p, LAYERS[stage - 10][4]), dtype=np.float16)
        elif stage == 30:
            out = np.empty((n, EMBED_DIM), dtype=np.float16)
        else:
            raise ValueError("unknown stage")
        check(self._lib.frhot_debug_read(self._h, stage, n, C.c_void_p(out.ctypes.data), out.nbytes))
        return out

    def activation_peaks(self, n: int):
        """Range check after an embed of n crops: (largest stored magnitude per stage, all finite?)
        for conv1_1 .. conv5_3 and fc6, in the fp16 units the library computes in (input_scale x
        the network's activations).  A stage that overflowed fp16 holds inf; later layers can turn
        it into NaN and ReLU turns NaN into 0, so an out-of-range model does NOT always show in
        its descriptors: run this on representative crops after loading new weights and lower
        input_scale until every peak stays well below 65504."""
        peaks, finite = [], True
        for stage in list(range(10, 23)) + [30]:
            a = self.debug_read(stage, n).astype(np.float32)
            ok = np.isfinite(a)
            finite = finite and bool(ok.all())
            peaks.append(float(np.abs(a[ok]).max()) if ok.any() else float("inf"))
        return peaks, finite

    def timing(self, enable: bool = True):
        check(self._lib.frhot_ctx_timing(self._h, 1 if enable else 0))

    def stage_ms(self) -> dict:
        ms = np.zeros(5, dtype=np.float32)
        check(self._lib.frhot_ctx_stage_ms(self._h, _ptr(ms, C.c_float)))
        return dict(zip(("preprocess", "conv", "fc", "scan", "merge"), (float(x) for x in ms)))

    def merge_topk_dev(self, scores_dev: int, rows_dev: int, p: int, bq: int, k: int, out_s_dev: int, out_r_dev: int):
        check(self._lib.frhot_merge_topk_dev(self._h, C.c_void_p(scores_dev), C.c_void_p(rows_dev), p, bq, k,
                                             C.c_void_p(out_s_dev), C.c_void_p(out_r_dev)))

    def merge_topk_packed_dev(self, packed_dev: int, shard_stride_bytes: int, p: int, bq: int, k: int,
                              out_s_dev: int, out_r_dev: int):
        """Merge p shards' answers straight out of an all-gathered buffer: shard i's block at
        packed_dev + i * shard_stride_bytes = rows int64 [bq,k], then scores float32 [bq,k]."""
        check(self._lib.frhot_merge_topk_packed_dev(self._h, C.c_void_p(packed_dev), shard_stride_bytes, p, bq, k,
                                                    C.c_void_p(out_s_dev), C.c_void_p(out_r_dev)))

    @staticmethod
    def dist_unique_id() -> bytes:
        """128 bytes that rank 0 makes and every rank passes to dist_init (carried by the host's own means)."""
        buf = C.create_string_buffer(128)
        check(_lib.load().frhot_dist_unique_id(buf))
        return buf.raw

    def dist_init(self, rank: int, world: int, unique_id: bytes):
        """Join a communicator of `world` ranks (collective): the exchange of Gallery.search_sharded_dev
        then runs inside the library over RCCL."""
        if len(unique_id) != 128:
            raise ValueError("the id is 128 bytes")
        check(self._lib.frhot_dist_init(self._h, rank, world, C.c_char_p(unique_id)))

    def dist_shutdown(self):
        check(self._lib.frhot_dist_shutdown(self._h))

    def recognise(self, gallery: "Gallery", images, qkeys=None, k: int = 5, gamma: bool = False, flip: bool = True,
                  want_embeddings: bool = False, want_sharpness: bool = False):
        """Fused embed -> search.  Returns (scores [n,k], rows [n,k], embeddings or None), and with
        `want_sharpness` a fourth item: the crops' Laplacian variances (see crop_sharpness)."""
        n = len(images)
        if n > self.max_batch:
            raise ValueError(f"batch {n} exceeds max_batch {self.max_batch}")
        arrs, ptrs, hw, strides = _crop_args(images)
        scores = np.empty((n, k), dtype=np.float32)
        rows = np.empty((n, k), dtype=np.int64)
        emb = np.empty((n, EMBED_DIM), dtype=np.float32) if want_embeddings else None
        qk = None if qkeys is None else np.ascontiguousarray(qkeys, dtype=np.int32)
        args = (self._h, gallery.handle, ptrs, _ptr(hw, C.c_int32), _ptr(strides, C.c_int32),
                _ptr(qk, C.c_int32) if qk is not None else None, n, self.flags(gamma, flip), k,
                _ptr(emb, C.c_float) if emb is not None else None, _ptr(scores, C.c_float), _ptr(rows, C.c_int64))
        if not want_sharpness:
            check(self._lib.frhot_recognise(*args))
            return scores, rows, emb
        sums = np.empty((n, 2), dtype=np.int64)
        check(self._lib.frhot_recognise_gated(*args, _ptr(sums, C.c_int64)))
        return scores, rows, emb, laplacian_variance(sums, hw)

    def crop_sharpness_sums(self, images) -> np.ndarray:
        """int64 [n, 2]: sum and sum of squares of the 3x3 Laplacian of every crop's grey image."""
        n = len(images)
        out = np.empty((n, 2), dtype=np.int64)
        for lo in range(0, n, self.max_batch):
            chunk = images[lo:lo + self.max_batch]
            arrs, ptrs, hw, strides = _crop_args(chunk)
            sub = out[lo:lo + len(chunk)]
            check(self._lib.frhot_crop_sharpness(self._h, ptrs, _ptr(hw, C.c_int32), _ptr(strides, C.c_int32),
                                                 len(chunk), _ptr(sub, C.c_int64)))
        return out

    def crop_sharpness_dev(self, crops_dev: int, offsets: np.ndarray, hw: np.ndarray, strides: np.ndarray,
                           out_dev: int):
        """Device-pointer form (arguments as embed_dev); out_dev: int64 [n, 2].  Enqueues only."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        hw = np.ascontiguousarray(hw, dtype=np.int32)
        strides = np.ascontiguousarray(strides, dtype=np.int32)
        check(self._lib.frhot_crop_sharpness_dev(self._h, C.c_void_p(crops_dev), _ptr(offsets, C.c_int64),
                                                 _ptr(hw, C.c_int32), _ptr(strides, C.c_int32), len(offsets),
                                                 C.c_void_p(out_dev)))

    def crop_sharpness(self, images) -> np.ndarray:
        """float64 [n]: cv2.Laplacian(cv2.cvtColor(crop, COLOR_BGR2GRAY), CV_64F).var() of uint8 BGR
        crops, the measure of the reference's blur gate (app/src/utils/legacy.py:125-129)."""
        hw = np.array([np.asarray(im).shape[:2] for im in images], dtype=np.int64).reshape(-1, 2)
        return laplacian_variance(self.crop_sharpness_sums(images), hw)


class Gallery:
    """Device-resident [capacity, D] fp16 unit rows + int32 keys (frhot_gallery)."""

    def __init__(self, ctx: Context, dim: int, capacity: int):
        self._lib = ctx._lib
        self.ctx = ctx
        self.dim = dim
        self.capacity = capacity
        self._h = C.c_void_p()
        check(self._lib.frhot_gallery_create(ctx.handle, dim, capacity, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value and self.ctx.handle.value:
            self._lib.frhot_gallery_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def rows(self) -> int:
        return int(self._lib.frhot_gallery_rows(self._h))

    def upsert(self, vecs: np.ndarray, keys, rows):
        vecs = np.ascontiguousarray(vecs, dtype=np.float32).reshape(-1, self.dim)
        n = vecs.shape[0]
        keys = np.ascontiguousarray(np.broadcast_to(np.asarray(keys, dtype=np.int32), (n,)))
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(n)
        check(self._lib.frhot_gallery_upsert(self._h, _ptr(vecs, C.c_float), _ptr(keys, C.c_int32),
                                             _ptr(rows, C.c_int64), n))

    SEARCH_AUTO, SEARCH_BY_KEY, SEARCH_WHOLE = 0, 1, 2

    def search_mode(self, mode: int):
        """How searches whose queries carry keys look for rows: by key (only the 256-row tiles that
        may hold a key are scanned: the reference searches one collection's
        own storage, qdrant_database_FE/app.py:316-337), over the whole gallery, or whichever a cost
        estimate prefers (the default).  The answers are the same."""
        check(self._lib.frhot_gallery_search_mode(self._h, int(mode)))

    def tombstone(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        check(self._lib.frhot_gallery_tombstone(self._h, _ptr(rows, C.c_int64), rows.size))

    def fill_synthetic(self, seed: int, row0: int, n: int, nkeys: int = 0, dup_every: int = 0):
        check(self._lib.frhot_gallery_fill_synthetic(self._h, seed, row0, n, nkeys, dup_every))

    def read(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(-1)
        out = np.empty((rows.size, self.dim), dtype=np.float16)
        keys = np.empty(rows.size, dtype=np.int32)
        check(self._lib.frhot_gallery_read(self._h, _ptr(rows, C.c_int64), rows.size, C.c_void_p(out.ctypes.data),
                                           _ptr(keys, C.c_int32)))
        return out, keys

    def write_f16(self, rows_f16: np.ndarray, keys, rows):
        """Restore rows exactly as read(): fp16 unit rows [n, D], keys [n], destination rows [n]."""
        rows_f16 = np.ascontiguousarray(rows_f16, dtype=np.float16).reshape(-1, self.dim)
        n = rows_f16.shape[0]
        keys = np.ascontiguousarray(np.broadcast_to(np.asarray(keys, dtype=np.int32), (n,)))
        rows = np.ascontiguousarray(rows, dtype=np.int64).reshape(n)
        check(self._lib.frhot_gallery_write_f16(self._h, C.c_void_p(rows_f16.ctypes.data), _ptr(keys, C.c_int32),
                                                _ptr(rows, C.c_int64), n))

    def search(self, q: np.ndarray, qkeys=None, k: int = 5):
        """float32 queries [Bq, D] (host) -> (scores [Bq,k] f32 descending, rows [Bq,k] i64, -1 = none)."""
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, self.dim)
        bq = q.shape[0]
        qk = None if qkeys is None else np.ascontiguousarray(np.broadcast_to(np.asarray(qkeys, np.int32), (bq,)))
        scores = np.empty((bq, k), dtype=np.float32)
        rows = np.empty((bq, k), dtype=np.int64)
        check(self._lib.frhot_search(self._h, _ptr(q, C.c_float), _ptr(qk, C.c_int32) if qk is not None else None,
                                     bq, k, _ptr(scores, C.c_float), _ptr(rows, C.c_int64)))
        return scores, rows

    def search_dev(self, q_f16_dev: int, qkeys_dev: int | None, bq: int, k: int, scores_dev: int, rows_dev: int,
                   row_base: int = 0):
        check(self._lib.frhot_search_dev(self._h, C.c_void_p(q_f16_dev), C.c_void_p(qkeys_dev or 0), bq, k,
                                         C.c_void_p(scores_dev), C.c_void_p(rows_dev), row_base))

    def search_dev_keyed(self, q_f16_dev: int, qkeys, bq: int, k: int, scores_dev: int, rows_dev: int,
                         row_base: int = 0):
        """search_dev with the queries' keys as a host array (or None): the gallery is then searched
        by key where that pays (see search_mode).  Enqueues only."""
        qk = None if qkeys is None else np.ascontiguousarray(qkeys, dtype=np.int32).reshape(-1)
        if qk is not None and qk.size != bq:
            raise ValueError(f"{qk.size} keys for {bq} queries")
        check(self._lib.frhot_search_dev_keyed(self._h, C.c_void_p(q_f16_dev),
                                               _ptr(qk, C.c_int32) if qk is not None else None, bq, k,
                                               C.c_void_p(scores_dev), C.c_void_p(rows_dev), row_base))

    def search_dev_static(self, q_f16_dev: int, qkeys_dev: int | None, bq: int, k: int, scores_dev: int,
                          rows_dev: int, row_base: int = 0):
        """search_dev without any allocation: the scan's scratch is a static device array."""
        check(self._lib.frhot_search_dev_static(self._h, C.c_void_p(q_f16_dev), C.c_void_p(qkeys_dev or 0), bq, k,
                                                C.c_void_p(scores_dev), C.c_void_p(rows_dev), row_base))

    def search_sharded_dev(self, q_local_f16_dev: int, qkeys_local_dev: int | None, b: int, k: int, scores_dev: int,
                           rows_dev: int, row_base: int = 0):
        """This rank's b queries in, every rank's answers out ([world*b, k], rank-major, global rows):
        gather / scan of this shard / gather / merge, enqueued on the context stream (collective)."""
        check(self._lib.frhot_search_sharded_dev(self._h, C.c_void_p(q_local_f16_dev),
                                                 C.c_void_p(qkeys_local_dev or 0), b, k, C.c_void_p(scores_dev),
                                                 C.c_void_p(rows_dev), row_base))


__all__ = ["Context", "Gallery", "EMBED_DIM", "KEY_ANY"]
