#!/usr/bin/env python3
"""Developer tool: the fused call and the device-pointer call that search by key, against the
whole-gallery search, on random galleries and key sets.

  * frhot_recognise(crops, keys): by key == whole gallery, also with two callers in flight
    (each holds a staging slot with index arrays of its own);
  * frhot_search_dev_keyed(queries on the device, keys on the host, row base): == frhot_search_dev
    with the keys on the device, calls issued back to back without waiting in between.
Usage: python scripts/fuzz_recognise_keys.py [rounds] [seed]
"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from frhot import synth  # noqa: E402
from frhot.engine import Context, Gallery  # noqa: E402
from frhot.weights import pack_blob, synthetic_weights  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = Context(device=0, max_batch=48)
    ctx.load_model(pack_blob(synthetic_weights(seed=0xFACE)), input_scale=1.0 / 16.0)
    pool = synth.crops(seed=5, n=64, lo=100, hi=220)
    for r in range(rounds):
        d = int(rng.choice([512, 4096]))
        n = int(rng.choice([3_000, 60_000, 200_000])) if d == 512 else int(rng.choice([2_000, 30_000]))
        nkeys = int(rng.choice([2, 30, 300]))
        gal = Gallery(ctx, d, n)
        layout = int(rng.integers(0, 3))
        if layout == 0:       # keys strewn over every tile
            gal.fill_synthetic(seed=int(rng.integers(1, 1 << 30)), row0=0, n=n, nkeys=nkeys, dup_every=0)
        else:                 # collections stored one after the other
            gal.fill_synthetic(seed=int(rng.integers(1, 1 << 30)), row0=0, n=n, nkeys=-max(1, n // nkeys), dup_every=0)
        _, keys = gal.read(np.arange(n))
        dead = rng.integers(0, n, size=max(1, n // 50))
        gal.tombstone(dead)
        keys[dead] = -1
        live = np.flatnonzero(keys >= 0)

        def batch():
            b = int(rng.choice([1, 3, 17, 48]))
            imgs = [pool[i] for i in rng.integers(0, len(pool), size=b)]
            qk = keys[rng.choice(live, size=b)].copy()
            qk[rng.random(b) < 0.2] = -2
            qk[rng.random(b) < 0.1] = int(keys.max()) + 3
            return imgs, qk, int(rng.integers(1, 17))
        batches = [batch() for _ in range(4)]
        gal.search_mode(Gallery.SEARCH_WHOLE)
        want = [ctx.recognise(gal, im, qk, k)[:2] for im, qk, k in batches]
        gal.search_mode(Gallery.SEARCH_BY_KEY)
        got = [ctx.recognise(gal, im, qk, k)[:2] for im, qk, k in batches]
        with ThreadPoolExecutor(max_workers=2) as ex:
            got2 = list(ex.map(lambda a: ctx.recognise(gal, a[0], a[1], a[2])[:2], batches * 3))
        for i, w in enumerate(want):
            for g in (got[i], got2[i], got2[i + 4], got2[i + 8]):
                assert np.array_equal(g[1], w[1]) and np.array_equal(g[0], w[0]), f"round {r} batch {i}: recognise by key differs"
        # device-pointer form with a row base, several calls queued before anything is read
        outs = []
        for im, qk, k in batches:
            b = len(im)
            q = torch.randn((b, d), dtype=torch.float32)
            q = (q / q.norm(dim=1, keepdim=True)).to(torch.float16).cuda()
            qk_dev = torch.from_numpy(qk).cuda()
            s1 = torch.empty((b, k), dtype=torch.float32, device="cuda"); r1 = torch.empty((b, k), dtype=torch.int64, device="cuda")
            s2 = torch.empty_like(s1); r2 = torch.empty_like(r1)
            torch.cuda.synchronize()
            base = int(rng.integers(0, 1 << 40))
            gal.search_dev(q.data_ptr(), qk_dev.data_ptr(), b, k, s1.data_ptr(), r1.data_ptr(), base)
            gal.search_dev_keyed(q.data_ptr(), qk, b, k, s2.data_ptr(), r2.data_ptr(), base)
            outs.append((s1, r1, s2, r2, q, qk_dev))
        ctx.sync()
        for i, (s1, r1, s2, r2, _q, _k) in enumerate(outs):
            assert torch.equal(r1, r2) and torch.equal(s1, s2), f"round {r} batch {i}: search_dev_keyed differs"
        print(f"round {r}: d={d} n={n} keys={nkeys} layout={layout} ok", flush=True)
        gal.close()
    ctx.close()
    print("fuzz_recognise_keys: passed")


if __name__ == "__main__":
    main()
