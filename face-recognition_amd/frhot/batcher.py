"""Micro-batchers: turn concurrent single-item calls into GPU batches.

The reference embeds one crop per request inside a worker thread, up to 10 at a
time per process (app/src/services/face_service.py:42-43,231-233), and issues one
/search_point request per face, up to 10 in flight (face_service.py:42,286-291).
Here those callers enqueue their item and wait on a future; one dispatcher thread
per batcher drains its queue every `max_delay_s` (or as soon as `max_batch` items
wait) and makes a single device call over the batch: one frhot_embed for the crops
(EmbedBatcher), one gallery scan for the queries (SearchBatcher).  A scan costs the
same for 1 query as for 64, so grouping the searches matters as much as grouping
the crops.
"""
from __future__ import annotations

import queue
import threading
import time
from concurrent.futures import Future

import numpy as np


class EmbedBatcher:
    def __init__(self, ctx, max_batch: int | None = None, max_delay_s: float = 0.002):
        self.ctx = ctx
        self.max_batch = min(max_batch or ctx.max_batch, ctx.max_batch)
        self.max_delay_s = max_delay_s
        self._q: queue.Queue = queue.Queue()
        self._stop = threading.Event()
        self.batches = 0
        self.items = 0
        self._thread = threading.Thread(target=self._run, name="frhot-batcher", daemon=True)
        self._thread.start()

    def submit(self, crop: np.ndarray, gamma: bool = False) -> Future:
        """Queue one uint8 BGR crop; the future resolves to its float32 [4096] unit descriptor."""
        fut: Future = Future()
        self._q.put((crop, bool(gamma), fut))
        return fut

    def embed(self, crop: np.ndarray, gamma: bool = False, timeout: float | None = 60.0) -> np.ndarray:
        return self.submit(crop, gamma).result(timeout=timeout)

    def close(self):
        self._stop.set()
        self._q.put(None)
        self._thread.join(timeout=5.0)

    def _run(self):
        while not self._stop.is_set():
            item = self._q.get()
            if item is None:
                continue
            batch = [item]
            deadline = time.monotonic() + self.max_delay_s
            while len(batch) < self.max_batch:
                left = deadline - time.monotonic()
                if left <= 0:
                    break
                try:
                    nxt = self._q.get(timeout=left)
                except queue.Empty:
                    break
                if nxt is None:
                    break
                batch.append(nxt)
            # one device call per gamma setting (a flag of the whole batch)
            for flag in (False, True):
                group = [b for b in batch if b[1] == flag]
                if not group:
                    continue
                try:
                    out = self.ctx.embed([g[0] for g in group], gamma=flag)
                    for g, row in zip(group, out):
                        g[2].set_result(row.copy())
                except Exception as e:  # every waiter sees the failure
                    for g in group:
                        g[2].set_exception(e)
                self.batches += 1
                self.items += len(group)


class SearchBatcher:
    """Groups concurrent search_point calls into one GalleryService.search_points call."""

    def __init__(self, service, max_batch: int = 256, max_delay_s: float = 0.002):
        self.service = service
        self.max_batch = max_batch
        self.max_delay_s = max_delay_s
        self._q: queue.Queue = queue.Queue()
        self._stop = threading.Event()
        self.batches = 0
        self.items = 0
        self._thread = threading.Thread(target=self._run, name="frhot-search-batcher", daemon=True)
        self._thread.start()

    def submit(self, collection_name, vector_embedding, store_id) -> Future:
        """Queue one search; the future resolves to the service's (status_code, content)."""
        fut: Future = Future()
        self._q.put(((collection_name, vector_embedding, store_id), fut))
        return fut

    def search_point(self, collection_name, vector_embedding, store_id, timeout: float | None = 60.0):
        return self.submit(collection_name, vector_embedding, store_id).result(timeout=timeout)

    def close(self):
        self._stop.set()
        self._q.put(None)
        self._thread.join(timeout=5.0)

    def _run(self):
        while not self._stop.is_set():
            item = self._q.get()
            if item is None:
                continue
            batch = [item]
            deadline = time.monotonic() + self.max_delay_s
            while len(batch) < self.max_batch:
                left = deadline - time.monotonic()
                if left <= 0:
                    break
                try:
                    nxt = self._q.get(timeout=left)
                except queue.Empty:
                    break
                if nxt is None:
                    break
                batch.append(nxt)
            try:
                answers = self.service.search_points([b[0] for b in batch])
                for b, ans in zip(batch, answers):
                    b[1].set_result(ans)
            except Exception as e:  # every waiter sees the failure
                for b in batch:
                    if not b[1].done():
                        b[1].set_exception(e)
            self.batches += 1
            self.items += len(batch)
