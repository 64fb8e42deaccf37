#!/usr/bin/env python3
"""Developer experiment: do two independent embed+scan pipelines on one GPU (two contexts, two
streams, one host thread each) finish more faces per second than one?  Persistent kernels leave
CUs idle in a layer's last round; a second stream's kernels can start there.
Usage: python scripts/two_pipelines.py [npipe] [steps]
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "face-recognition_amd"))

import torch  # noqa: E402

from frhot import synth  # noqa: E402
from frhot.engine import Context, Gallery  # noqa: E402
from frhot.weights import pack_blob, synthetic_weights  # noqa: E402


def main():
    npipe = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    B, D, K, N = 256, 512, 5, 10_000_000
    dev = torch.device("cuda", 0)
    blob = pack_blob(synthetic_weights(seed=0xFACE))
    images = synth.crops(seed=0xFACE0003, n=B, lo=112, hi=448)
    buf, offsets, hw, strides = synth.pack_crops(images)
    pipes = []
    for p in range(npipe):
        ctx = Context(device=0, max_batch=B)
        ctx.load_model(blob, input_scale=1.0 / 16.0)
        gal = Gallery(ctx, D, N)
        gal.fill_synthetic(seed=1000 + p, row0=0, n=N, nkeys=0, dup_every=100)
        crops = torch.from_numpy(buf).to(dev)
        q = torch.empty((B, D), dtype=torch.float16, device=dev)
        s = torch.empty((B, K), dtype=torch.float32, device=dev)
        r = torch.empty((B, K), dtype=torch.int64, device=dev)
        pipes.append((ctx, gal, crops, q, s, r))
    torch.cuda.synchronize()
    flags = Context.flags(gamma=True, flip=True)

    def run(pipe, n):
        ctx, gal, crops, q, s, r = pipe
        for _ in range(n):
            ctx.embed_dev(crops.data_ptr(), offsets, hw, strides, flags, None, q.data_ptr(), D)
            gal.search_dev(q.data_ptr(), None, B, K, s.data_ptr(), r.data_ptr(), 0)
        ctx.sync()

    for pipe in pipes:
        run(pipe, 3)
    t0 = time.perf_counter()
    ts = [threading.Thread(target=run, args=(pipe, steps)) for pipe in pipes]
    [t.start() for t in ts]
    [t.join() for t in ts]
    dt = time.perf_counter() - t0
    print(f"{npipe} pipeline(s): {npipe * steps * B / dt:.0f} faces/s ({dt / steps * 1e3:.2f} ms per round of {npipe} x {B})")
    for ctx, gal, *_ in pipes:
        gal.close()
        ctx.close()


if __name__ == "__main__":
    main()
