#!/usr/bin/env python3
"""Developer tool: faces/s of frhot_recognise with HOST crops in and HOST answers out.

The headline bench keeps its crops resident in HBM; this script times the entry point the
reference's handlers would call (host numpy crops -> host scores/rows), which adds the
host-to-device copy of the crops and the device-to-host copy of the answers to every batch.
Usage: python scripts/host_path_rate.py [batch] [gallery_rows] [dim] [steps] [callers]
callers > 1: that many threads call at once (the reference's handlers arrive from up to 10 worker threads).
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "face-recognition_amd"))

from frhot import synth  # noqa: E402
from frhot.engine import Context, Gallery  # noqa: E402
from frhot.weights import pack_blob, synthetic_weights  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    callers = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    ctx = Context(device=0, max_batch=B)
    ctx.load_model(pack_blob(synthetic_weights(seed=0xFACE)), input_scale=1.0 / 16.0)
    gal = Gallery(ctx, D, N)
    gal.fill_synthetic(seed=0xFACE0000 + 1000, row0=0, n=N, nkeys=0, dup_every=100)
    images = synth.crops(seed=0xFACE0003, n=B, lo=112, hi=448)
    nbytes = sum(im.nbytes for im in images)
    for _ in range(2):
        ctx.recognise(gal, images, None, 5, gamma=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        s, r, _ = ctx.recognise(gal, images, None, 5, gamma=True)
    dt = (time.perf_counter() - t0) / steps
    print(f"host path: batch {B}, {nbytes / 1e6:.1f} MB of crops per batch, gallery {N}x{D}: "
          f"{dt * 1e3:.2f} ms per batch = {B / dt:.0f} faces/s (rows ok: {bool((r >= 0).all())})")
    if callers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=callers) as pool:
            list(pool.map(lambda _: ctx.recognise(gal, images, None, 5, gamma=True), range(callers)))
            t0 = time.perf_counter()
            list(pool.map(lambda _: ctx.recognise(gal, images, None, 5, gamma=True), range(steps * callers)))
            dt = (time.perf_counter() - t0) / (steps * callers)
        print(f"host path, {callers} callers in flight: {dt * 1e3:.2f} ms per batch = {B / dt:.0f} faces/s")
    gal.close()
    ctx.close()


if __name__ == "__main__":
    main()
