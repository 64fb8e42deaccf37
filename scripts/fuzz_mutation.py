#!/usr/bin/env python3
"""Developer tool: random gallery traffic (upserts that move rows between keys, tombstones, restored
fp16 rows, synthetic fills) with searches in between; every search is answered by key and over the
whole gallery and the two answers must be identical (the per-key tile lists the library keeps on the
host have to stay a superset of where each key's rows are), and every few rounds the answer is also
checked against the CPU oracle on a host mirror of the gallery.
Usage: python scripts/fuzz_mutation.py [rounds] [seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

from frhot.engine import Context, Gallery  # noqa: E402
from oracle import search as osr  # noqa: E402


def unit(x):
    return x / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-30)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = Context(device=0, max_batch=4)
    for trial in range(3):
        d = int(rng.choice([64, 128, 512]))
        cap = int(rng.choice([3_000, 40_000, 300_000]))
        nkeys = int(rng.choice([3, 40, 400]))
        gal = Gallery(ctx, d, cap)
        mirror = np.zeros((cap, d), np.float16)
        keys = np.full(cap, -1, np.int32)
        top = 0                                          # rows the scan covers
        for r in range(rounds):
            op = int(rng.integers(0, 6))
            if op <= 1 or top == 0:                      # upsert a run or a scatter of rows
                n = int(rng.integers(1, max(2, cap // 8)))
                if rng.integers(0, 2):
                    lo = int(rng.integers(0, cap - n + 1))
                    rows = np.arange(lo, lo + n)
                else:
                    rows = rng.choice(cap, size=n, replace=False)
                v = rng.standard_normal((n, d), dtype=np.float32)
                k = rng.integers(0, nkeys, size=n).astype(np.int32)
                if rng.integers(0, 3) == 0:
                    k[:] = int(rng.integers(0, nkeys))   # a whole run of one key
                gal.upsert(v, k, rows)
                mirror[rows], back = gal.read(rows)      # the device normalises in its own arithmetic: take what it stored
                assert np.array_equal(back, k)
                assert np.abs(mirror[rows].astype(np.float32) - unit(v)).max() <= 1e-3
                keys[rows] = k
                top = max(top, int(rows.max()) + 1)
            elif op == 2:                                # tombstone
                rows = rng.integers(0, top, size=int(rng.integers(1, max(2, top // 20))))
                gal.tombstone(rows)
                keys[rows] = -1
            elif op == 3:                                # restore stored rows elsewhere (snapshot path)
                n = int(rng.integers(1, max(2, min(top, cap // 10))))
                src = rng.integers(0, top, size=n)
                dst = rng.choice(cap, size=n, replace=False)
                got16, gotk = gal.read(src)
                assert np.array_equal(got16, mirror[src]) and np.array_equal(gotk, keys[src]), "read differs from mirror"
                newk = np.where(gotk < 0, -1, (gotk + 1) % nkeys).astype(np.int32)
                gal.write_f16(got16, newk, dst)
                mirror[dst] = got16
                keys[dst] = newk
                top = max(top, int(dst.max()) + 1)
            elif op == 4 and cap >= 40_000:              # synthetic fill of a region (keys made on the device)
                n = int(rng.integers(256, cap // 4))
                lo = int(rng.integers(0, cap - n + 1))
                nk = int(rng.choice([nkeys, -int(rng.integers(50, 2000))]))
                gal.fill_synthetic(seed=int(rng.integers(1, 1 << 30)), row0=lo, n=n, nkeys=nk, dup_every=0)
                top = max(top, lo + n)
                got16, gotk = gal.read(np.arange(lo, lo + n))
                mirror[lo:lo + n] = got16
                keys[lo:lo + n] = gotk
            # search
            bq = int(rng.choice([1, 5, 64, 70, 200]))
            k = int(rng.integers(1, 17))
            live = np.flatnonzero(keys[:top] >= 0)
            if len(live) == 0:
                continue
            t = rng.choice(live, size=bq)
            q = mirror[t].astype(np.float32) + 0.3 * rng.standard_normal((bq, d), dtype=np.float32) / np.sqrt(d)
            qk = keys[t].copy()
            qk[rng.random(bq) < 0.15] = -2
            qk[rng.random(bq) < 0.05] = int(keys.max()) + 7           # a key no row has
            gal.search_mode(Gallery.SEARCH_WHOLE)
            s_w, r_w = gal.search(q, qk, k)
            gal.search_mode(Gallery.SEARCH_BY_KEY)
            s_k, r_k = gal.search(q, qk, k)
            assert np.array_equal(r_w, r_k) and np.array_equal(s_w, s_k), f"trial {trial} round {r}: by key != whole gallery"
            if r % 5 == 0 and top * bq <= 3e7:
                ref_s, ref_r = osr.cosine_topk(q, mirror[:top].astype(np.float32), k, keys=keys[:top], qkeys=qk,
                                               gallery_dtype=np.float16)
                assert np.array_equal(r_w >= 0, ref_r >= 0), f"trial {trial} round {r}: presence differs from the oracle"
                with np.errstate(invalid="ignore"):      # empty slots hold -inf on both sides
                    diff = np.where(r_w >= 0, s_w - ref_s, 0.0)
                assert np.abs(diff).max() <= 3e-5, f"trial {trial} round {r}: scores differ"
        print(f"trial {trial}: d={d} capacity={cap} keys={nkeys}: {rounds} rounds ok", flush=True)
        gal.close()
    ctx.close()
    print("fuzz_mutation: passed")


if __name__ == "__main__":
    main()
