"""In-process replacement of the reference's DB service (qdrant_database_FE/app.py).

Same operations, same status codes and JSON contents, but the points live in one
device-resident gallery (frhot.engine.Gallery) instead of Qdrant collections:

  * a collection name plus the payload's store_id intern to one int32 key per row,
    so the reference's "collection + must store_id == value" filter
    (qdrant_database_FE/app.py:316-327) is one key comparison inside the scan;
  * the row -> (point id, payload) table, the <=10-faces-per-id eviction
    (app.py:207-240), delete-by-payload-id (app.py:409-438) and the threshold + vote
    (app.py:340-384) stay on the host;
  * per-collection snapshots (app.py:130-196) are one .npz file each: the stored fp16
    rows read back from the device plus the point table, restored bit for bit.

Methods return (status_code, content) exactly as the handlers answer.
"""
from __future__ import annotations

import datetime
import json
import os
import threading
import uuid

import numpy as np

from .vote import vote

DIM = 4096
SEARCH_LIMIT = 5


TILE = 256      # rows per gallery tile: what one scan step covers and a by-key search skips or reads


class _RowAllocator:
    """Gallery rows handed out so that a key's rows sit together.  The reference keeps every
    collection in storage of its own (qdrant_database_FE/app.py:108-128); here a key (collection,
    store) takes whole 256-row tiles while unclaimed tiles last, so a search for that key reads only
    its own tiles.  When every tile is claimed, rows are borrowed from the key with the most free
    rows: still correct (the library lists every tile a key was written to), only less compact."""

    def __init__(self, capacity: int):
        self.capacity = int(capacity)
        self._ntiles = (self.capacity + TILE - 1) // TILE
        self._next_tile = 0
        self._owner: dict[int, int] = {}            # tile -> key that claimed it
        self._free: dict[int, list[int]] = {}       # key -> free rows in its tiles (taken from the end)

    def free_count(self) -> int:
        unclaimed = max(0, min(self.capacity, self._ntiles * TILE) - self._next_tile * TILE)
        return unclaimed + sum(len(v) for v in self._free.values())

    def alloc(self, key: int) -> int:
        mine = self._free.get(key)
        if mine:
            return mine.pop()
        if self._next_tile < self._ntiles:
            t = self._next_tile
            self._next_tile += 1
            self._owner[t] = key
            lo, hi = t * TILE, min((t + 1) * TILE, self.capacity)
            self._free[key] = list(range(hi - 1, lo - 1, -1))      # ascending rows come off the end
            return self._free[key].pop()
        donor = max(self._free.items(), key=lambda kv: len(kv[1]), default=(None, None))[1]
        if not donor:
            raise MemoryError("gallery capacity exhausted")
        return donor.pop(0)                                        # the donor's highest row

    def release(self, row: int):
        self._free.setdefault(self._owner[row // TILE], []).append(row)


class GalleryService:
    def __init__(self, ctx, capacity: int = 1 << 16, dim: int = DIM, gallery=None):
        """`gallery`: an object with the upsert / tombstone / search / close methods of
        frhot.engine.Gallery to use instead of creating one on `ctx` (tests inject a stand-in)."""
        self.dim = dim
        self.capacity = capacity
        if gallery is None:
            from .engine import Gallery
            gallery = Gallery(ctx, dim, capacity)
        self.gallery = gallery
        self._lock = threading.RLock()
        self._held = threading.local()             # .depth: how often the calling thread holds _lock through _locked()
        self._collections: dict[str, None] = {}
        self._keys: dict[tuple, int] = {}          # (collection, store_id) -> device key
        self._points: dict[int, dict] = {}         # row -> {"uid", "payload", "collection", "seq"}
        self._by_id: dict[tuple, list] = {}        # (collection, payload id) -> rows in insertion order
        self._rows = _RowAllocator(capacity)
        self._seq = 0

    def close(self):
        self.gallery.close()

    class _Locked:
        def __init__(self, svc):
            self.svc = svc

        def __enter__(self):
            self.svc._lock.acquire()
            self.svc._held.depth = getattr(self.svc._held, "depth", 0) + 1

        def __exit__(self, *exc):
            self.svc._held.depth -= 1
            self.svc._lock.release()
            return False

    def _locked(self):
        """The table lock, counting per thread how deep the holder is in."""
        return self._Locked(self)

    # ------------------------------------------------------------ collections
    def get_collections(self):
        """app.py:100-106."""
        with self._lock:
            return {"status": 200, "collections": list(self._collections)}

    def collection_exists(self, name) -> bool:
        with self._lock:
            return name in self._collections

    def create_collection(self, collection_name):
        """app.py:108-128."""
        if collection_name is None:
            return 404, {"message": "Collection name not found!"}
        with self._lock:
            if collection_name not in self._collections:
                self._collections[collection_name] = None
                return 201, {"message": "Collection created"}
            return 200, {"message": "Collection existed"}

    def point_count(self, collection_name) -> int:
        with self._lock:
            return sum(1 for p in self._points.values() if p["collection"] == collection_name)

    # ----------------------------------------------------------------- points
    def _key(self, collection, store_id) -> int:
        k = (collection, store_id)
        if k not in self._keys:
            self._keys[k] = len(self._keys)
        return self._keys[k]

    def _alloc_row(self, key: int) -> int:
        return self._rows.alloc(key)

    def _drop_rows(self, rows):
        if not rows:
            return
        self.gallery.tombstone(np.asarray(rows, dtype=np.int64))
        for r in rows:
            p = self._points.pop(r)
            lst = self._by_id[(p["collection"], p["payload"]["id"])]
            lst.remove(r)
            if not lst:
                del self._by_id[(p["collection"], p["payload"]["id"])]
            self._rows.release(r)

    def _enforce_face_limit(self, collection, pid, limit=10):
        """app.py:207-240: among the first limit+5 points of this id, delete the oldest so limit-1 stay."""
        rows = list(self._by_id.get((collection, pid), []))[:limit + 5]
        if len(rows) >= limit:
            rows.sort(key=lambda r: (self._points[r]["payload"].get("time_created", ""), self._points[r]["seq"]))
            self._drop_rows(rows[:len(rows) - limit + 1])

    def insert_point(self, collection_name, vector_embedding, id, name, store_id, is_update_id=False,
                     time_created=None, uid=None):
        """app.py:242-297.  `uid`: the point's id (a fresh uuid4, app.py:288, unless the caller brings
        one: replicas of this table that apply the same insert must agree on it)."""
        if time_created is None:
            time_created = datetime.datetime.now().strftime("%Y/%m/%d %H:%M:%S")
        if collection_name == "" or collection_name is None:
            return 404, {"message": "Collection name not found or invalid!"}
        with self._lock:
            if collection_name not in self._collections:
                return 404, {"message": "Collection name not found or invalid!"}
            if vector_embedding is None or len(vector_embedding) != self.dim:
                return 404, {"message": "Embedding not found or invalid!"}
            if id is None or id == "":
                return 404, {"message": "ID not found or invalid!"}
            if name is None or id == "":
                return 404, {"message": "Name not found or invalid!"}
            try:
                if not is_update_id:
                    self._enforce_face_limit(collection_name, id, limit=10)
            except Exception as e:  # the reference only logs this failure (app.py:271-272)
                print(f"Error checking face limit: {e}")
            payload = {'id': id, 'name': name, 'store_id': store_id, 'time_created': time_created}
            try:
                vec = np.asarray(vector_embedding, dtype=np.float32).reshape(1, self.dim)
                if is_update_id:
                    rows = self._by_id.get((collection_name, id))
                    if not rows:
                        return 404, {"message": "ID not found!"}
                    row = rows[0]
                    old = self._points[row]
                    self.gallery.upsert(vec, self._key(collection_name, old["payload"].get("store_id")), [row])
                    return 201, {"message": "Point inserted"}
                key = self._key(collection_name, store_id)
                row = self._alloc_row(key)
                try:
                    self.gallery.upsert(vec, key, [row])
                except Exception:
                    self._rows.release(row)         # the row was never written: hand it back
                    raise
                self._seq += 1
                self._points[row] = {"uid": uid or str(uuid.uuid4()), "payload": payload, "collection": collection_name,
                                     "seq": self._seq}
                self._by_id.setdefault((collection_name, id), []).append(row)
                return 201, {"message": "Point inserted"}
            except Exception as e:
                return 404, {"message": str(e)}

    def delete_point(self, collection_name, id):
        """app.py:409-438."""
        if collection_name is None or collection_name == "":
            return 404, {"message": "Collection name not found or invalid!"}
        with self._lock:
            if collection_name not in self._collections:
                return 404, {"message": "Collection name not found or invalid!"}
            if id is None or id == "":
                return 404, {"message": "ID not found or invalid!"}
            try:
                self._drop_rows(list(self._by_id.get((collection_name, id), [])))
                return 200, {"message": "Point deleted"}
            except Exception as e:
                return 404, {"message": str(e)}

    # ----------------------------------------------------------------- search
    def _validate_search(self, collection_name, vector_embedding):
        if collection_name is None or collection_name == "":
            return 404, {"message": "Collection name not found or invalid!"}
        if collection_name not in self._collections:
            return 404, {"message": "Collection name not found or invalid!"}
        if vector_embedding is None or len(vector_embedding) != self.dim:
            return 404, {"message": "Embedding not found or invalid!"}
        return None

    def _hits(self, scores, rows, seq0=None):
        """Payloads of the rows a scan returned.  `seq0`: the write sequence number when the scan was
        handed to the device without the table lock held; a row written after that may have been
        scored with its previous occupant's vector and is left out (the search then counts as having
        run between that occupant's deletion and the new point's insertion)."""
        out = []
        for s, r in zip(scores, rows):
            p = self._points.get(int(r)) if r >= 0 else None
            if p is not None and (seq0 is None or p["seq"] <= seq0):
                out.append((float(s), p["payload"]))
        return out

    # The table lock is NOT held while the device works on a search: enrolments and deletions of other
    # requests go on meanwhile (the library orders their device writes with the scan on the context's
    # stream), where qdrant_database_FE serves its handlers concurrently too.  A service whose search is
    # a collective keeps the lock: every rank has to make its calls in the same order.
    _search_holds_lock = False

    class _Unlocked:
        def __init__(self, lock, active):
            self.lock, self.active = lock, active

        def __enter__(self):
            if self.active:
                self.lock.release()

        def __exit__(self, *exc):
            if self.active:
                self.lock.acquire()
            return False

    def _device_call(self):
        """`with self._lock: ... with self._device_call(): <device work>`: the lock is given up for
        the inner block (only when it is held exactly once, i.e. not from inside another locked call)."""
        once = not self._search_holds_lock and getattr(self._held, "depth", 0) == 1
        return self._Unlocked(self._lock, once)

    def search_point(self, collection_name, vector_embedding, store_id):
        """app.py:300-386."""
        return self.search_points([(collection_name, vector_embedding, store_id)])[0]

    def search_points(self, requests):
        """Batched form: [(collection_name, vector, store_id)] -> [(status, content)], one device scan."""
        with self._locked():
            out = [None] * len(requests)
            idx, vecs, keys = [], [], []
            for i, (col, vec, store_id) in enumerate(requests):
                bad = self._validate_search(col, vec)
                if bad is not None:
                    out[i] = bad
                    continue
                k = self._keys.get((col, store_id))
                if k is None:           # nothing was ever stored under this collection + store
                    out[i] = (200, {"message": "Point not found", "data": []})
                    continue
                idx.append(i)
                vecs.append(np.asarray(vec, dtype=np.float32))
                keys.append(k)
            if idx or self._always_search:
                try:
                    seq0 = self._seq
                    with self._device_call():
                        scores, rows = self._search_batch(vecs, keys)
                    for j, i in enumerate(idx):
                        out[i] = (200, vote(self._hits(scores[j], rows[j], seq0)))
                except Exception as e:
                    for i in idx:
                        out[i] = (404, {"message": str(e)})
            return out

    _always_search = False   # (a sharded gallery's search is a collective: every rank makes the call, with or without queries)

    def _search_batch(self, vecs, keys):
        return self.gallery.search(np.stack(vecs), np.asarray(keys, np.int32), SEARCH_LIMIT)

    NO_KEY = 0x7FFFFFFF      # a query key no row carries: the crop is still embedded, nothing matches

    def recognise_points(self, requests, want_embeddings: bool = False, gamma: bool = True,
                         want_sharpness: bool = False):
        """Fused form of get_embedding + search_point for a batch: [(collection_name, crop uint8
        [h,w,3] BGR, store_id)] -> [(status, content, descriptor or None)] from ONE frhot_recognise
        call, so the descriptors go from the network to the scan without leaving HBM
        (face_service.py:231-233,275-291 and app.py:300-386 in one step).  `gamma`: apply the
        gamma-1.5 LUT of face_service.py:223 on the device.  Same validation, threshold and vote
        as search_point; needs a gallery created on a frhot.engine.Context.  `want_sharpness`: every
        answer carries a fourth item, the crop's Laplacian variance before the LUT (the measure of
        the blur gate, legacy.py:125-129), from the same device call."""
        if not requests:
            return []
        with self._locked():
            out = [None] * len(requests)
            keys, bad = [], {}
            for i, (col, _crop, store_id) in enumerate(requests):
                if col is None or col == "" or col not in self._collections:
                    bad[i] = (404, {"message": "Collection name not found or invalid!"})
                k = self._keys.get((col, store_id))
                keys.append(self.NO_KEY if (k is None or i in bad) else k)
            seq0 = self._seq
            with self._device_call():
                scores, rows, emb, *sharp = self.gallery.ctx.recognise(
                    self.gallery, [r[1] for r in requests], np.asarray(keys, np.int32), SEARCH_LIMIT, gamma=gamma,
                    want_embeddings=want_embeddings, want_sharpness=want_sharpness)
            for i in range(len(requests)):
                e = emb[i].copy() if emb is not None else None
                if i in bad:
                    out[i] = bad[i] + (e,)
                elif keys[i] == self.NO_KEY:
                    out[i] = (200, {"message": "Point not found", "data": []}, e)
                else:
                    out[i] = (200, vote(self._hits(scores[i], rows[i], seq0)), e)
                if sharp:
                    out[i] += (float(sharp[0][i]),)
            return out

    # -------------------------------------------------------------- snapshots
    # One file per snapshot, <dir>/<collection>/<collection>-<timestamp>.frsnap.npz, holding
    # "rows" fp16 [n, dim] exactly as stored on the device and "points" = JSON list of
    # {uid, payload, seq}.  Same life cycle as the reference's Qdrant snapshots.
    def _snap_dir(self, directory, collection_name):
        return os.path.join(directory, collection_name)

    def list_snapshots(self, collection_name, directory):
        """app.py:143-146: the snapshots of a collection, newest first."""
        d = self._snap_dir(directory, collection_name)
        if not os.path.isdir(d):
            return []
        out = []
        for fn in sorted(os.listdir(d), reverse=True):
            if fn.endswith(".frsnap.npz"):
                st = os.stat(os.path.join(d, fn))
                out.append({"name": fn, "size": st.st_size,
                            "creation_time": datetime.datetime.fromtimestamp(st.st_mtime).isoformat()})
        return out

    def create_snapshot(self, collection_name, directory):
        """app.py:130-141: delete the collection's old snapshots, then write a new one."""
        with self._lock:
            if collection_name not in self._collections:
                return 404, {"message": "Collection name not found or invalid!"}
            d = self._snap_dir(directory, collection_name)
            os.makedirs(d, exist_ok=True)
            for old in self.list_snapshots(collection_name, directory):
                os.remove(os.path.join(d, old["name"]))
            rows = sorted(r for r, p in self._points.items() if p["collection"] == collection_name)
            if rows:
                vec, _keys = self.gallery.read(np.asarray(rows, dtype=np.int64))
            else:
                vec = np.zeros((0, self.dim), np.float16)
            points = [{"uid": self._points[r]["uid"], "payload": self._points[r]["payload"],
                       "seq": self._points[r]["seq"]} for r in rows]
            self._seq += 1
            name = f"{collection_name}-{datetime.datetime.now().strftime('%Y-%m-%d-%H-%M-%S')}-{self._seq}.frsnap.npz"
            path = os.path.join(d, name)
            with open(path, "wb") as f:
                np.savez(f, rows=np.asarray(vec, np.float16), dim=np.int64(self.dim),
                         points=np.frombuffer(json.dumps(points).encode("utf-8"), dtype=np.uint8))
            return 200, {"name": name, "size": os.path.getsize(path), "points": len(rows)}

    def recover_snapshot(self, collection_name, path):
        """app.py:171-196: (create the collection if missing and) replace its points by the snapshot's."""
        try:
            parts = collection_name.split("_")
            if len(parts) < 2 or parts[1] not in ("Employees", "Customers"):
                return 404, {"message": "Collection name not found or invalid!"}
            with np.load(path, allow_pickle=False) as z:
                vec = np.asarray(z["rows"], np.float16)
                if int(z["dim"]) != self.dim or vec.ndim != 2 or vec.shape[1] != self.dim:
                    return 404, {"message": "snapshot dimension does not match this gallery"}
                points = json.loads(bytes(z["points"]).decode("utf-8"))
            if len(points) != vec.shape[0]:
                return 404, {"message": "snapshot is inconsistent"}
            with self._lock:
                # The snapshot's rows are written to fresh rows FIRST and the collection's current
                # points dropped only after that succeeded: a snapshot that does not fit, or a
                # failing device write, leaves the live collection as it was and leaks no row.
                old_rows = [r for r, p in self._points.items() if p["collection"] == collection_name]
                room = self._rows.free_count()
                if len(points) > room + len(old_rows):
                    return 404, {"message": f"snapshot holds {len(points)} points, the gallery has room for "
                                            f"{room + len(old_rows)}"}
                if len(points) > room:
                    # it only fits in the rows the collection occupies now: those go first, and a
                    # failing write then leaves the collection empty (its rows are handed back)
                    self._drop_rows(old_rows)
                    old_rows = []
                existed = collection_name in self._collections
                rows, keys = [], []
                try:
                    for pt in points:
                        keys.append(self._key(collection_name, pt["payload"].get("store_id")))
                        rows.append(self._alloc_row(keys[-1]))
                    if rows:
                        self.gallery.write_f16(vec, np.asarray(keys, np.int32), np.asarray(rows, np.int64))
                except Exception:
                    for r in rows:
                        self._rows.release(r)
                    if rows:
                        try:        # whatever part of the write landed must not answer searches
                            self.gallery.tombstone(np.asarray(rows, dtype=np.int64))
                        except Exception:
                            pass
                    if not existed:
                        self._collections.pop(collection_name, None)
                    raise
                self._collections.setdefault(collection_name, None)
                self._drop_rows(old_rows)
                for r, pt in zip(rows, points):
                    self._seq = max(self._seq, int(pt["seq"]))
                    self._points[r] = {"uid": pt["uid"], "payload": pt["payload"], "collection": collection_name,
                                       "seq": int(pt["seq"])}
                    self._by_id.setdefault((collection_name, pt["payload"]["id"]), []).append(r)
            return 200, {"message": "Recover snapshot successfully"}
        except Exception as e:
            return 404, {"message": str(e)}
