#!/usr/bin/env python3
"""Developer tool: randomised parity sweep of the gallery scan against the CPU oracle.

Random gallery sizes (1 row to millions), dimensions, query counts, k, key filters with tombstones,
planted exact duplicates and queries that tie; every case is checked like the GPU tests check
theirs (identical rows except inside score ties within 2e-5, scores within 2e-5).
Usage: python scripts/fuzz_search.py [cases] [seed] [first case to run]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

from frhot import synth  # noqa: E402
from frhot.engine import Context, Gallery  # noqa: E402
from oracle import search as osr  # noqa: E402
from test_gpu_hotpath import _check_topk  # noqa: E402


DIMS = [64, 64, 128, 192, 512]      # FUZZ_DIMS=64,4096 on the command line's environment replaces them


def one_case(ctx, rng, case, skip=False):
    d = int(rng.choice(DIMS))
    kind = rng.integers(0, 10)
    if kind == 0:
        n = int(rng.integers(1, 300))
    elif d == 4096:
        n = int(rng.integers(300, 40_000))
    elif kind < 6:
        n = int(rng.integers(300, 120_000))
    elif kind < 9:
        n = int(rng.integers(120_000, 600_000)) if d <= 128 else int(rng.integers(50_000, 150_000))
    else:
        d = 64
        n = int(rng.integers(2_100_000, 2_600_000))        # a block walks more than 32 tiles: bounds on
    bq = int(rng.choice([1, 3, 17, 64, 65, 128, 200, 256, 300, 600, 2100])) if n < 700_000 else int(rng.integers(1, 64))
    if bq > 600 and n * bq > 3e8:
        bq = 600
    if d == 4096 and bq > 300:
        bq = 300
    k = int(rng.integers(1, 17))
    with_keys = bool(rng.integers(0, 2))
    g = rng.standard_normal((n, d), dtype=np.float32)
    g /= np.linalg.norm(g, axis=1, keepdims=True)
    ndup = min(200, n)
    dup = rng.integers(0, n, size=(ndup, 2))
    g[dup[:, 0]] = g[dup[:, 1]]
    targets = rng.integers(0, n, size=bq)
    q = synth.planted_queries(rng, g[targets], float(rng.choice([0.5, 0.7, 0.95])))
    q[-1] = g[dup[0, 1]]
    keys = rng.integers(0, int(rng.choice([2, 3, 50])), size=n).astype(np.int32) if with_keys else None
    qkeys = None
    # every random draw of the case happens before the device is used, so that a later case can be
    # replayed alone (`first` on the command line) from the same seed
    dead = rng.integers(0, n, size=max(1, n // 100)) if with_keys else None
    mode = int(rng.integers(0, 3))
    if skip:
        return n, d, bq, k, with_keys
    print(f"case {case}: n={n} d={d} bq={bq} k={k} keys={with_keys} mode={mode} ...", flush=True)
    gal = Gallery(ctx, d, n)
    try:
        for lo in range(0, n, 200_000):
            hi = min(n, lo + 200_000)
            gal.upsert(g[lo:hi], keys[lo:hi] if with_keys else 0, np.arange(lo, hi))
        if with_keys:
            qkeys = keys[targets].copy()
            qkeys[::5] = -2
            gal.tombstone(dead)
            keys = keys.copy()
            keys[dead] = -1
        gal.search_mode(mode)                         # cost rule / by key / whole gallery: the same answer
        got_s, got_r = gal.search(q, qkeys, k)
        ref_s, ref_r = osr.cosine_topk(q, g, k, keys=keys, qkeys=qkeys, gallery_dtype=np.float16)
        qn = osr.unit_rows(q).astype(np.float16).astype(np.float32)

        def score(i, row):
            return float(qn[i] @ osr.unit_rows(g[row:row + 1]).astype(np.float16).astype(np.float32)[0])
        try:
            _check_topk(got_s, got_r, ref_s, ref_r, score, tol=2e-5)
        except AssertionError as e:
            import re
            m = re.search(r"query (\d+)", str(e))
            if m:
                i = int(m.group(1))
                qk = None if qkeys is None else int(qkeys[i])
                live = None if keys is None else int((keys == qk).sum()) if qk is not None and qk >= 0 else int((keys >= 0).sum())
                print(f"FAILED query {i}: key {qk}, live rows with it {live}\n got rows {got_r[i].tolist()}\n got scores "
                      f"{got_s[i].tolist()}\n ref rows {ref_r[i].tolist()}\n ref scores {ref_s[i].tolist()}", flush=True)
                if keys is not None:
                    print(" keys of got rows", [int(keys[r]) if r >= 0 else None for r in got_r[i]], flush=True)
            raise
    finally:
        gal.close()
    return n, d, bq, k, with_keys


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if os.environ.get("FUZZ_DIMS"):
        DIMS[:] = [int(x) for x in os.environ["FUZZ_DIMS"].split(",")]
    rng = np.random.default_rng(seed)
    ctx = Context(device=0, max_batch=4)
    t0 = time.time()
    for c in range(cases):
        desc = one_case(ctx, rng, c, skip=c < first)
        if c < first:
            continue
        print(f"case {c}: n={desc[0]} d={desc[1]} bq={desc[2]} k={desc[3]} keys={desc[4]} ok ({time.time() - t0:.0f} s)", flush=True)
    ctx.close()
    print(f"fuzz_search: {cases} cases passed")


if __name__ == "__main__":
    main()
