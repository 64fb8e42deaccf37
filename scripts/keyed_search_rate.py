#!/usr/bin/env python3
"""Developer tool: what a search costs when the gallery holds many collections (keys).

The reference searches one collection's own storage (qdrant_database_FE/app.py:316-337); the
gallery here is one table, and a search whose queries carry keys scans only the 256-row tiles
that may hold each key.  This times frhot_search (host queries in, host answers out) over the
whole gallery against by key, for collections stored one after the other.
Usage: python scripts/keyed_search_rate.py [rows] [dim] [rows_per_key]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "face-recognition_amd"))

import numpy as np  # noqa: E402

from frhot.engine import Context, Gallery  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    run = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    ctx = Context(device=0, max_batch=16)
    gal = Gallery(ctx, d, n)
    gal.fill_synthetic(seed=11, row0=0, n=n, nkeys=-run, dup_every=0)
    nkeys = (n + run - 1) // run
    rng = np.random.default_rng(3)
    print(f"gallery {n} x {d}, {nkeys} keys in runs of {run} rows")
    for bq, distinct in ((1, 1), (8, 1), (8, 8), (64, 4), (64, 64), (256, 32)):
        ks = rng.choice(nkeys, size=distinct, replace=False)
        qkeys = ks[rng.integers(0, distinct, size=bq)].astype(np.int32)
        rows = qkeys.astype(np.int64) * run + rng.integers(0, run, size=bq)
        rows = np.minimum(rows, n - 1)
        q16, rk = gal.read(rows)
        assert (rk == qkeys).all()
        q = q16.astype(np.float32)
        res = {}
        for name, mode in (("whole", Gallery.SEARCH_WHOLE), ("by key", Gallery.SEARCH_BY_KEY), ("auto", Gallery.SEARCH_AUTO)):
            gal.search_mode(mode)
            for _ in range(3):
                s, r = gal.search(q, qkeys, 5)
            t0 = time.perf_counter()
            it = 30
            for _ in range(it):
                s, r = gal.search(q, qkeys, 5)
            res[name] = ((time.perf_counter() - t0) / it * 1e3, s, r)
        same = all(np.array_equal(res["whole"][2], res[m][2]) and np.array_equal(res["whole"][1], res[m][1])
                   for m in ("by key", "auto"))
        assert (res["by key"][2][:, 0] == rows).all()
        print(f"  {bq:3d} queries over {distinct:2d} keys: whole gallery {res['whole'][0]:7.3f} ms   by key {res['by key'][0]:7.3f} ms"
              f"   auto {res['auto'][0]:7.3f} ms   ({res['whole'][0] / res['by key'][0]:.1f}x)   same answers: {same}")
    gal.close()
    ctx.close()


if __name__ == "__main__":
    main()
