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


class _MicroBatcher:
    """Queue + dispatcher thread: collects items for at most `max_delay_s` (or until `max_batch`
    wait) and hands the batch to `_dispatch`, which resolves every item's future."""

    def __init__(self, max_batch: int, max_delay_s: float, name: str):
        self.max_batch = max_batch
        self.max_delay_s = max_delay_s
        self._q: queue.Queue = queue.Queue()
        self._stop = threading.Event()
        self.batches = 0
        self.items = 0
        self._thread = threading.Thread(target=self._run, name=name, daemon=True)
        self._thread.start()

    def _put(self, payload) -> Future:
        fut: Future = Future()
        self._q.put((payload, fut))
        return fut

    def close(self):
        self._stop.set()
        self._q.put(None)
        self._thread.join(timeout=5.0)

    def _dispatch(self, batch):          # batch: list of (payload, future)
        raise NotImplementedError

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
                self._dispatch(batch)
            except Exception as e:  # every waiter sees the failure
                for _, fut in batch:
                    if not fut.done():
                        fut.set_exception(e)


MAX_CROP_SIDE = 16384     # frhot_embed refuses larger crops


def crop_problem(crop) -> str | None:
    """Why frhot_embed would refuse this crop (None: it is usable).  Mirrors the library's own
    checks so that a bad crop fails its own request instead of the whole device batch:
    uint8 [h, w, 3], sides within 1..16384, and an aspect fit into 224 x 224 that keeps at
    least one row and one column (DeepFace's resize: dsize = int(side * min(224/h, 224/w)))."""
    a = np.asarray(crop)
    if a.ndim != 3 or a.shape[2] != 3:
        return f"crop must have shape [h, w, 3], not {a.shape}"
    if a.dtype != np.uint8:
        return f"crop must be uint8, not {a.dtype}"
    h, w = a.shape[:2]
    if not (1 <= h <= MAX_CROP_SIDE and 1 <= w <= MAX_CROP_SIDE):
        return f"crop sides {h} x {w} outside 1..{MAX_CROP_SIDE}"
    f = min(224.0 / h, 224.0 / w)
    if int(w * f) < 1 or int(h * f) < 1:
        return f"crop {h} x {w} resizes to an empty image"
    return None


class EmbedBatcher(_MicroBatcher):
    """Groups concurrent single-crop embeds into one frhot_embed per gamma setting."""

    def __init__(self, ctx, max_batch: int | None = None, max_delay_s: float = 0.002):
        self.ctx = ctx
        super().__init__(min(max_batch or ctx.max_batch, ctx.max_batch), max_delay_s, "frhot-batcher")

    def submit(self, crop: np.ndarray, gamma: bool = False) -> Future:
        """Queue one uint8 BGR crop; the future resolves to its float32 [4096] unit descriptor."""
        bad = crop_problem(crop)
        if bad is not None:             # fails this request only, before it can join a batch
            fut: Future = Future()
            fut.set_exception(ValueError(bad))
            return fut
        return self._put((crop, bool(gamma)))

    def embed(self, crop: np.ndarray, gamma: bool = False, timeout: float | None = 60.0) -> np.ndarray:
        return self.submit(crop, gamma).result(timeout=timeout)

    def _dispatch(self, batch):
        # one device call per gamma setting (a flag of the whole batch)
        for flag in (False, True):
            group = [b for b in batch if b[0][1] == flag]
            if not group:
                continue
            try:
                out = self.ctx.embed([g[0][0] for g in group], gamma=flag)
                for g, row in zip(group, out):
                    g[1].set_result(row.copy())
                self.batches += 1
            except Exception as e:
                # the library refuses a call as a whole: find the crop(s) it refuses by embedding
                # one at a time, so that every other request of the batch still gets its answer
                if len(group) == 1:
                    group[0][1].set_exception(e)
                    self.batches += 1
                else:
                    for g in group:
                        try:
                            g[1].set_result(self.ctx.embed([g[0][0]], gamma=flag)[0].copy())
                        except Exception as e1:
                            g[1].set_exception(e1)
                        self.batches += 1
            self.items += len(group)


class DetectBatcher(_MicroBatcher):
    """Groups concurrent single-frame detections (the reference's detect_face and anti-spoofing pass
    run once per request, legacy.py:60-65,80-86) into one frhot_det_detect call: one frame takes the
    detector 1.15 ms, eight frames 1.6 ms together."""

    def __init__(self, detector, max_batch: int | None = None, max_delay_s: float = 0.002, conf: float = 0.25):
        self.detector = detector
        self.conf = conf
        super().__init__(min(max_batch or detector.max_batch, detector.max_batch), max_delay_s, "frhot-det-batcher")

    def submit(self, frame: np.ndarray) -> Future:
        """Queue one uint8 BGR frame; the future resolves to its float32 [n, 20] detections
        (Detector.detect's rows)."""
        a = np.asarray(frame)
        if a.ndim != 3 or a.shape[2] != 3 or a.dtype != np.uint8:       # fails this request only, before it can join a batch
            fut: Future = Future()
            fut.set_exception(ValueError(f"frame must be uint8 [h, w, 3], not {a.dtype} {a.shape}"))
            return fut
        return self._put(a)

    def detect(self, frame: np.ndarray, timeout: float | None = 60.0) -> np.ndarray:
        return self.submit(frame).result(timeout=timeout)

    def extract_faces(self, frame, expand_percentage: int = 10, anti_spoof=None, assume_real: bool = False):
        """Detector.as_extract_faces's entries for one frame, detected as part of a batch."""
        from .detector import faces_of
        if anti_spoof is None and not assume_real:
            raise ValueError("no anti-spoofing model: pass anti_spoof=callable, or assume_real=True to accept every face")
        frame = np.ascontiguousarray(frame)
        faces = faces_of(frame, self.detect(frame), expand_percentage)
        for f in faces:
            f["is_real"] = bool(anti_spoof(f["face"])) if anti_spoof is not None else True
        return faces

    def _dispatch(self, batch):
        try:
            out = self.detector.detect([b[0] for b in batch], conf=self.conf)
            for b, rows in zip(batch, out):
                b[1].set_result(rows)
            self.batches += 1
        except Exception as e:
            if len(batch) == 1:
                batch[0][1].set_exception(e)
                self.batches += 1
            else:       # find the frame(s) the library refuses; every other request still gets its answer
                for b in batch:
                    try:
                        b[1].set_result(self.detector.detect([b[0]], conf=self.conf)[0])
                    except Exception as e1:
                        b[1].set_exception(e1)
                    self.batches += 1
        self.items += len(batch)


class SearchBatcher(_MicroBatcher):
    """Groups concurrent search_point calls into one GalleryService.search_points call."""

    def __init__(self, service, max_batch: int = 256, max_delay_s: float = 0.002):
        self.service = service
        super().__init__(max_batch, max_delay_s, "frhot-search-batcher")

    def submit(self, collection_name, vector_embedding, store_id) -> Future:
        """Queue one search; the future resolves to the service's (status_code, content)."""
        return self._put((collection_name, vector_embedding, store_id))

    def search_point(self, collection_name, vector_embedding, store_id, timeout: float | None = 60.0):
        return self.submit(collection_name, vector_embedding, store_id).result(timeout=timeout)

    def _dispatch(self, batch):
        answers = self.service.search_points([b[0] for b in batch])
        for b, ans in zip(batch, answers):
            b[1].set_result(ans)
        self.batches += 1
        self.items += len(batch)


class RecogniseBatcher(_MicroBatcher):
    """Groups concurrent recognise / enrol requests into one GalleryService.recognise_points call
    (one frhot_recognise: crops in, voted answers out, descriptors stay in HBM)."""

    def __init__(self, service, max_batch: int, max_delay_s: float = 0.002):
        self.service = service
        super().__init__(max_batch, max_delay_s, "frhot-recognise-batcher")

    def submit(self, collection_name, crop: np.ndarray, store_id, want_embedding: bool = False,
               want_sharpness: bool = False) -> Future:
        """Queue one crop (uint8 BGR, not gamma-adjusted: the device applies the LUT); the future
        resolves to (status_code, content, descriptor or None), with `want_sharpness` followed by
        the crop's Laplacian variance (the blur gate's measure, taken before the LUT)."""
        bad = crop_problem(crop)
        if bad is not None:
            fut: Future = Future()
            fut.set_exception(ValueError(bad))
            return fut
        return self._put((collection_name, crop, store_id, bool(want_embedding), bool(want_sharpness)))

    def _dispatch(self, batch):
        reqs = [b[0][:3] for b in batch]
        want = any(b[0][3] for b in batch)
        sharp = any(b[0][4] for b in batch)
        try:
            answers = self.service.recognise_points(reqs, want_embeddings=want, want_sharpness=sharp)
            self.batches += 1
        except Exception as e:
            if len(batch) == 1:
                batch[0][1].set_exception(e)
                self.batches += 1
                self.items += 1
                return
            answers = []
            for b in batch:         # one at a time: only the request the library refuses fails
                try:
                    answers.append(self.service.recognise_points([b[0][:3]], want_embeddings=b[0][3],
                                                                 want_sharpness=sharp)[0])
                except Exception as e1:
                    answers.append(e1)
                self.batches += 1
        for b, ans in zip(batch, answers):
            if isinstance(ans, Exception):
                b[1].set_exception(ans)
            else:
                b[1].set_result(ans[:2] + (ans[2] if b[0][3] else None,) + (ans[3:4] if b[0][4] else ()))
        self.items += len(batch)
