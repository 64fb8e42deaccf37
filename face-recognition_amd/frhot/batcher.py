"""Micro-batcher: turns concurrent single-crop calls into GPU batches.

The reference embeds one crop per request inside a worker thread, up to 10 at a
time per process (app/src/services/face_service.py:42-43,231-233).  Here those
threads enqueue their crop and wait on a future; one dispatcher thread drains the
queue every `max_delay_s` (or as soon as `max_batch` crops wait) and runs a single
frhot_embed over the batch.
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
