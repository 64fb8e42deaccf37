"""Seeded synthetic inputs for tests and benchmarks (numpy only, no GPU).

There is no network for datasets or checkpoints, so crops, galleries and
queries are generated: crops are smooth uint8 BGR images (low-pass noise plus a
gradient, so the bilinear resize matters) with sizes drawn from a range.
"""
from __future__ import annotations

import numpy as np


def crop(rng: np.random.Generator, h: int, w: int) -> np.ndarray:
    """One smooth uint8 BGR image of h x w."""
    ch, cw = max(2, h // 16 + 2), max(2, w // 16 + 2)
    coarse = rng.uniform(0.0, 255.0, size=(ch, cw, 3))
    ys = np.linspace(0, ch - 1, h)
    xs = np.linspace(0, cw - 1, w)
    y0 = np.floor(ys).astype(int).clip(0, ch - 2)
    x0 = np.floor(xs).astype(int).clip(0, cw - 2)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    a = coarse[y0][:, x0]
    b = coarse[y0][:, x0 + 1]
    c = coarse[y0 + 1][:, x0]
    d = coarse[y0 + 1][:, x0 + 1]
    img = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    gy = np.linspace(-20.0, 20.0, h)[:, None, None]
    img = img + gy + rng.normal(0.0, 3.0, size=(h, w, 3))
    # fancy indexing leaves the sum in a transposed memory order: hand out plain row-major pixels
    return np.ascontiguousarray(np.clip(np.rint(img), 0, 255).astype(np.uint8))


def crops(seed: int, n: int, lo: int = 112, hi: int = 448) -> list:
    """n crops with independent heights and widths in [lo, hi]."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        h = int(rng.integers(lo, hi + 1))
        w = int(rng.integers(lo, hi + 1))
        out.append(crop(rng, h, w))
    return out


def pack_crops(images: list):
    """Tightly pack HxWx3 uint8 images: (buffer uint8, offsets int64, hw int32 [n,2], strides int32)."""
    offsets, hw, strides = [], [], []
    total = 0
    for im in images:
        assert im.dtype == np.uint8 and im.ndim == 3 and im.shape[2] == 3
        offsets.append(total)
        hw.append((im.shape[0], im.shape[1]))
        strides.append(3 * im.shape[1])
        total += (im.shape[0] * im.shape[1] * 3 + 15) & ~15
    buf = np.zeros(total, dtype=np.uint8)
    for im, off in zip(images, offsets):
        buf[off:off + im.size] = np.ascontiguousarray(im).reshape(-1)
    return buf, np.asarray(offsets, np.int64), np.asarray(hw, np.int32), np.asarray(strides, np.int32)


def unit_gallery(seed: int, n: int, d: int, sparse: bool = False) -> np.ndarray:
    """Unit-norm float32 rows: gaussian, or |gaussian| * Bernoulli(0.3) like a post-ReLU descriptor."""
    rng = np.random.default_rng(seed)
    g = rng.standard_normal(size=(n, d), dtype=np.float32)
    if sparse:
        g = np.abs(g) * (rng.random(size=(n, d), dtype=np.float32) < 0.3)
        g[:, 0] += 1e-3
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    return g.astype(np.float32)


def planted_queries(rng: np.random.Generator, rows: np.ndarray, target_cos: float = 0.8) -> np.ndarray:
    """Queries = gallery rows plus gaussian noise scaled so cos(query, row) ~ target_cos."""
    rows = rows.astype(np.float32)
    d = rows.shape[1]
    sigma = np.sqrt((1.0 / target_cos ** 2 - 1.0) / d)
    q = rows + rng.standard_normal(size=rows.shape, dtype=np.float32) * np.float32(sigma)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)
