"""Oracle: exact cosine top-k with a collection/store filter, threshold + id vote.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference lines restated here:
  * search call ......... qdrant_database_FE/app.py:316-337 (exact=True, limit=5,
                          Distance.COSINE from :116, must store_id == value)
  * threshold + vote .... qdrant_database_FE/app.py:42-43,340-384
  * insert / evict ...... qdrant_database_FE/app.py:207-297
  * delete .............. qdrant_database_FE/app.py:409-438
Third-party behaviour restated (qdrant v1.14.1, PARITY UNPINNED): a COSINE
collection stores g/||g|| in f32 (left unchanged when the norm is ~0),
normalises the query the same way and scores by f32 dot product; exact=True
is a full scan, so the answer is the true top-`limit` under the filter, in
descending score; the order inside exact score ties is unspecified (this
oracle breaks ties by ascending row so tests are deterministic).
"""
from __future__ import annotations

import numpy as np

THRESHOLD_PASS = 0.54
THRESHOLD_SEARCH = 0.54
ANY_KEY = -2        # query key that matches every live row
TOMBSTONE = -1      # row key of a deleted row; matches nothing


def unit_rows(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    n = np.sqrt((x.astype(np.float64) ** 2).sum(axis=-1, keepdims=True)).astype(np.float32)
    n = np.where(n < np.float32(1e-30), np.float32(1.0), n)
    return (x / n).astype(np.float32)


def cosine_topk(q, g, k, keys=None, qkeys=None, gallery_dtype=np.float32, chunk=65536):
    """Exact top-k.  q [Bq,D], g [N,D] -> (scores f32 [Bq,k], rows i64 [Bq,k], -1 padded).

    gallery_dtype=np.float16 scores against the fp16-rounded unit gallery and
    query (the storage format of the HIP library) with f32 accumulation;
    np.float32 is the reference's own precision.
    """
    qn = unit_rows(q).astype(gallery_dtype).astype(np.float32)
    gn = unit_rows(g).astype(gallery_dtype).astype(np.float32)
    Bq, N = qn.shape[0], gn.shape[0]
    best_s = np.full((Bq, 0), -np.inf, dtype=np.float32)
    best_r = np.zeros((Bq, 0), dtype=np.int64)
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        s = qn @ gn[lo:hi].T
        if keys is not None:
            kk = np.asarray(keys[lo:hi])[None, :]
            qk = np.asarray(qkeys)[:, None]
            ok = np.where(qk == ANY_KEY, kk >= 0, kk == qk)
            s = np.where(ok, s, -np.inf).astype(np.float32)
        r = np.broadcast_to(np.arange(lo, hi, dtype=np.int64)[None, :], s.shape)
        cs = np.concatenate([best_s, s], axis=1)
        cr = np.concatenate([best_r, r], axis=1)
        order = np.lexsort((cr, -cs), axis=1)[:, :k]
        best_s = np.take_along_axis(cs, order, axis=1)
        best_r = np.take_along_axis(cr, order, axis=1)
    if best_s.shape[1] < k:
        pad = k - best_s.shape[1]
        best_s = np.concatenate([best_s, np.full((Bq, pad), -np.inf, np.float32)], axis=1)
        best_r = np.concatenate([best_r, np.full((Bq, pad), -1, np.int64)], axis=1)
    best_r = np.where(np.isneginf(best_s), -1, best_r)
    return best_s.astype(np.float32), best_r


def _kth_largest(s: np.ndarray, k: int) -> np.ndarray:
    """Per row of s, its k-th largest value (threaded torch.topk when torch is importable)."""
    try:
        import torch
        return torch.topk(torch.from_numpy(np.ascontiguousarray(s)), k, dim=1).values[:, -1].numpy()
    except ImportError:
        n = s.shape[1]
        return np.partition(s, n - k, axis=1)[:, n - k]


def topk_stored(q16, g16, k, keys=None, qkeys=None, chunk=65536, row0=0):
    """Exact top-k over rows ALREADY in the library's storage format: q16 [Bq,D] and g16 [N,D]
    are fp16 (or fp16-valued float32) and are scored as they stand, products and sums in
    float32; nothing is normalised.  Same answer rules as cosine_topk (score descending, ties
    by ascending row, -1 / -inf padding, key filter), but sized for millions of rows: per
    chunk only the rows at or above the chunk's k-th best score (all ties included) are kept
    and sorted.  `row0` is added to the returned rows (a slice of a larger gallery)."""
    qn = np.ascontiguousarray(q16, dtype=np.float32)
    Bq, N = qn.shape[0], g16.shape[0]
    cand_s = [[] for _ in range(Bq)]
    cand_r = [[] for _ in range(Bq)]
    qk = None if keys is None else np.asarray(qkeys)[:, None]
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        s = qn @ np.asarray(g16[lo:hi], dtype=np.float32).T
        if keys is not None:
            kk = np.asarray(keys[lo:hi])[None, :]
            ok = np.where(qk == ANY_KEY, kk >= 0, kk == qk)
            s = np.where(ok, s, np.float32(-np.inf))
        kk_ = min(k, hi - lo)
        thr = _kth_largest(s, kk_)
        for i in range(Bq):
            idx = np.nonzero((s[i] >= thr[i]) & np.isfinite(s[i]))[0]
            cand_s[i].append(s[i, idx])
            cand_r[i].append(idx.astype(np.int64) + lo)
    out_s = np.full((Bq, k), -np.inf, dtype=np.float32)
    out_r = np.full((Bq, k), -1, dtype=np.int64)
    for i in range(Bq):
        cs = np.concatenate(cand_s[i]) if cand_s[i] else np.zeros(0, np.float32)
        cr = np.concatenate(cand_r[i]) if cand_r[i] else np.zeros(0, np.int64)
        order = np.lexsort((cr, -cs))[:k]
        out_s[i, :len(order)] = cs[order]
        out_r[i, :len(order)] = cr[order] + row0
    return out_s, out_r


def cosine_topk_fast(q, g_unit, k):
    """Timing-oriented variant for the CPU baseline: g_unit is already unit float32 (as a
    server would keep it); one BLAS product per chunk, argpartition, then an exact sort of
    the k survivors.  Same scores as cosine_topk; tie order inside exact ties may differ."""
    qn = unit_rows(q)
    Bq, N = qn.shape[0], g_unit.shape[0]
    best_s = np.full((Bq, 0), -np.inf, dtype=np.float32)
    best_r = np.zeros((Bq, 0), dtype=np.int64)
    chunk = 1 << 18
    for lo in range(0, N, chunk):
        s = qn @ g_unit[lo:lo + chunk].T
        kk = min(k, s.shape[1])
        idx = np.argpartition(-s, kk - 1, axis=1)[:, :kk]
        cs = np.concatenate([best_s, np.take_along_axis(s, idx, axis=1)], axis=1)
        cr = np.concatenate([best_r, idx.astype(np.int64) + lo], axis=1)
        order = np.lexsort((cr, -cs), axis=1)[:, :k]
        best_s = np.take_along_axis(cs, order, axis=1)
        best_r = np.take_along_axis(cr, order, axis=1)
    return best_s.astype(np.float32), best_r


def vote(result):
    """qdrant_database_FE/app.py:340-384, literally.

    `result` is the scored-point list of the search, descending score: a list of
    (score, payload) with payload a dict holding 'id'.  Returns the JSON content
    the handler answers with (status 200): {"message", "data"}.
    """
    data = [(s, p) for (s, p) in result if s >= THRESHOLD_PASS]
    if len(data) == 0:
        return {"message": "Point not found", "data": []}
    data_dict = {}
    for s, p in result:
        if s >= THRESHOLD_PASS:
            if p['id'] in data_dict:
                data_dict[p['id']] += 1
            else:
                data_dict[p['id']] = 1
    if len(set(data_dict.values())) == 1 and len(data_dict) > 1:
        data_result = [(s, p) for (s, p) in result if s >= THRESHOLD_SEARCH]
        if len(data_result) == 0:
            return {"message": "Point not found", "data": []}
        i_max = max(data_result, key=lambda x: x[0])
        return {"message": "Point found", "data": [(i_max[0], i_max[1])]}
    id_max = max(data_dict, key=data_dict.get)
    scores = []
    for s, p in result:
        if p['id'] == id_max:
            scores.append(s)
    score = max(scores)
    payload = None
    for s, p in result:
        if p['id'] == id_max:
            payload = p
            break
    return {"message": "Point found", "data": [(score, payload)]}


class Collections:
    """In-memory restatement of the DB facade's point operations on Qdrant.

    Points are (uuid-like int, unit f32 vector, payload).  Only the behaviour the
    handlers rely on is modelled: create, insert with the <=10-per-id eviction,
    delete by payload id, exact filtered search + vote.
    """

    def __init__(self, dim=4096):
        self.dim = dim
        self.cols = {}
        self._next = 0

    def create_collection(self, name):
        """app.py:108-128 -> (status, content)."""
        if name is None:
            return 404, {"message": "Collection name not found!"}
        if name not in self.cols:
            self.cols[name] = []
            return 201, {"message": "Collection created"}
        return 200, {"message": "Collection existed"}

    def _enforce_face_limit(self, name, pid, limit=10):
        """app.py:207-240: scroll limit+5 points with this payload id, drop the oldest."""
        pts = [p for p in self.cols[name] if p[2]['id'] == pid][:limit + 5]
        if len(pts) >= limit:
            pts_sorted = sorted(pts, key=lambda p: p[2].get('time_created', ''))
            n_del = len(pts) - limit + 1
            dead = {p[0] for p in pts_sorted[:n_del]}
            self.cols[name] = [p for p in self.cols[name] if p[0] not in dead]

    def insert_point(self, name, vec, pid, pname, store_id, time_created, is_update_id=False):
        """app.py:242-297 -> (status, content)."""
        if name == "" or name is None or name not in self.cols:
            return 404, {"message": "Collection name not found or invalid!"}
        if vec is None or len(vec) != self.dim:
            return 404, {"message": "Embedding not found or invalid!"}
        if pid is None or pid == "":
            return 404, {"message": "ID not found or invalid!"}
        if pname is None or pid == "":
            return 404, {"message": "Name not found or invalid!"}
        if not is_update_id:
            self._enforce_face_limit(name, pid, 10)
        payload = {'id': pid, 'name': pname, 'store_id': store_id, 'time_created': time_created}
        v = unit_rows(np.asarray(vec, dtype=np.float32)[None])[0]
        if is_update_id:
            hit = [i for i, p in enumerate(self.cols[name]) if p[2]['id'] == pid]
            if not hit:
                return 404, {"message": "ID not found!"}
            old = self.cols[name][hit[0]]
            self.cols[name][hit[0]] = (old[0], v, old[2])
            return 201, {"message": "Point inserted"}
        self._next += 1
        self.cols[name].append((self._next, v, payload))
        return 201, {"message": "Point inserted"}

    def delete_point(self, name, pid):
        """app.py:409-438 -> (status, content)."""
        if name is None or name == "" or name not in self.cols:
            return 404, {"message": "Collection name not found or invalid!"}
        if pid is None or pid == "":
            return 404, {"message": "ID not found or invalid!"}
        self.cols[name] = [p for p in self.cols[name] if p[2]['id'] != pid]
        return 200, {"message": "Point deleted"}

    def search_hits(self, name, vec, store_id, limit=5):
        """app.py:316-337: filtered exact top-`limit`, descending score."""
        pts = [p for p in self.cols[name] if p[2]['store_id'] == store_id]
        if not pts:
            return []
        g = np.stack([p[1] for p in pts])
        s, r = cosine_topk(np.asarray(vec, np.float32)[None], g, min(limit, len(pts)))
        return [(float(s[0, i]), pts[int(r[0, i])][2]) for i in range(s.shape[1]) if r[0, i] >= 0]

    def search_point(self, name, vec, store_id):
        """app.py:300-386 -> (status, content)."""
        if name is None or name == "" or name not in self.cols:
            return 404, {"message": "Collection name not found or invalid!"}
        if vec is None or len(vec) != self.dim:
            return 404, {"message": "Embedding not found or invalid!"}
        return 200, vote(self.search_hits(name, vec, store_id))
