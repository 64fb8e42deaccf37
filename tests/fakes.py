"""CPU stand-ins used by the host-logic tests (no GPU): a numpy gallery with the method
set of frhot.engine.Gallery, and an embedder with the method set of EmbedBatcher.
They exist only so that the Python host code can be exercised without a device; the
arithmetic they perform comes from the oracle."""
import hashlib

import numpy as np

from oracle import search as osr


class NumpyGallery:
    def __init__(self, dim, capacity):
        self.dim, self.capacity = dim, capacity
        self.g = np.zeros((capacity, dim), np.float32)
        self.keys = np.full(capacity, -1, np.int32)
        self.nrows = 0

    def upsert(self, vecs, keys, rows):
        vecs = np.asarray(vecs, np.float32).reshape(-1, self.dim)
        rows = np.asarray(rows, np.int64).reshape(-1)
        keys = np.broadcast_to(np.asarray(keys, np.int32), rows.shape)
        self.g[rows] = osr.unit_rows(vecs)
        self.keys[rows] = keys
        self.nrows = max(self.nrows, int(rows.max()) + 1)

    def tombstone(self, rows):
        self.keys[np.asarray(rows, np.int64)] = -1

    def read(self, rows):
        rows = np.asarray(rows, np.int64).reshape(-1)
        return self.g[rows].astype(np.float16), self.keys[rows].copy()

    def write_f16(self, rows_f16, keys, rows):
        rows = np.asarray(rows, np.int64).reshape(-1)
        self.g[rows] = np.asarray(rows_f16, np.float16).astype(np.float32).reshape(len(rows), self.dim)
        self.keys[rows] = np.broadcast_to(np.asarray(keys, np.int32), rows.shape)
        self.nrows = max(self.nrows, int(rows.max()) + 1)

    def search(self, q, qkeys=None, k=5):
        q = np.asarray(q, np.float32).reshape(-1, self.dim)
        if self.nrows == 0:
            return np.full((len(q), k), -np.inf, np.float32), np.full((len(q), k), -1, np.int64)
        qk = np.full(len(q), -2, np.int32) if qkeys is None else np.broadcast_to(np.asarray(qkeys, np.int32), (len(q),))
        return osr.cosine_topk(q, self.g[:self.nrows], k, keys=self.keys[:self.nrows], qkeys=qk)

    def close(self):
        pass


class HashEmbedder:
    """Deterministic pseudo-descriptor of an image: identical crops -> identical unit vectors,
    different crops -> nearly orthogonal ones."""

    def __init__(self, dim=4096):
        self.dim = dim
        self.calls = 0

    def embed(self, crop, gamma=False):
        self.calls += 1
        h = hashlib.sha256(np.ascontiguousarray(crop).tobytes() + bytes([int(gamma)])).digest()
        rng = np.random.default_rng(int.from_bytes(h[:8], "little"))
        v = rng.standard_normal(self.dim).astype(np.float32)
        return v / np.linalg.norm(v)


class FakeContext:
    """Stand-in for frhot.engine.Context.recognise: HashEmbedder descriptors searched in a NumpyGallery."""

    def __init__(self, max_batch=16, dim=4096):
        self.max_batch = max_batch
        self.embedder = HashEmbedder(dim)
        self.calls = []

    def recognise(self, gallery, images, qkeys=None, k=5, gamma=False, flip=True, want_embeddings=False,
                  want_sharpness=False):
        self.calls.append(len(images))
        emb = np.stack([self.embedder.embed(im, gamma=gamma) for im in images])
        s, r = gallery.search(emb, qkeys, k)
        out = (s, r, (emb if want_embeddings else None))
        return out + (self.crop_sharpness(images),) if want_sharpness else out

    def crop_sharpness(self, images):
        from oracle import quality as oq
        return np.array([oq.laplacian_var(im) for im in images], np.float64)
