"""Oracle: numpy restatement of the library's deterministic synthetic gallery.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  This has no counterpart in
the reference; it restates frhot_gallery_fill_synthetic (include/frhot.h) so
that tests can check the device fill bit for bit: value(row, col) = sum of the
four low bytes of splitmix64(seed + row*D + col) minus 510, the row is divided
by the IEEE float32 square root of its exact integer sum of squares, and each
product is rounded once to float32 and once to fp16.
"""
from __future__ import annotations

import numpy as np

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def source_rows(rows: np.ndarray, dup_every: int) -> np.ndarray:
    rows = np.asarray(rows, dtype=np.int64)
    if dup_every > 0:
        dup = (rows > 0) & (rows % dup_every == dup_every - 1)
        return np.where(dup, rows - 1, rows)
    return rows


def rows_f16(seed: int, rows, d: int, dup_every: int = 0) -> np.ndarray:
    """The fp16 rows the device fill stores for global row indices `rows`."""
    src = source_rows(rows, dup_every).astype(np.uint64)
    with np.errstate(over="ignore"):
        idx = np.uint64(seed) + src[:, None] * np.uint64(d) + np.arange(d, dtype=np.uint64)[None, :]
    h = splitmix64(idx)
    v = ((h & np.uint64(255)) + ((h >> np.uint64(8)) & np.uint64(255)) + ((h >> np.uint64(16)) & np.uint64(255))
         + ((h >> np.uint64(24)) & np.uint64(255))).astype(np.int64) - 510
    ss = (v * v).sum(axis=1).astype(np.uint64)
    nrm = np.sqrt(ss.astype(np.float32))
    inv = np.where(ss == 0, np.float32(1.0), np.float32(1.0) / nrm).astype(np.float32)
    return (v.astype(np.float32) * inv[:, None]).astype(np.float32).astype(np.float16)


def keys(seed: int, rows, nkeys: int, dup_every: int = 0) -> np.ndarray:
    src = source_rows(rows, dup_every).astype(np.uint64)
    if nkeys < 0:                      # runs of -nkeys consecutive rows per key
        return (src // np.uint64(-nkeys)).astype(np.int32)
    if nkeys == 0:
        return np.zeros(len(src), dtype=np.int32)
    with np.errstate(over="ignore"):
        h = splitmix64(np.uint64(seed) ^ (src * np.uint64(0xD1B54A32D192ED03)))
    return (h % np.uint64(nkeys)).astype(np.int32)
