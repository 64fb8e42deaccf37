#!/usr/bin/env python3
"""Developer tool: the blur gate's Laplacian sums on random crop geometries (1 pixel to ~2000 a side,
widths around the 256-pixel strip and the 4-pixel load, row-strided and column-band views, flat,
noisy and smooth content) against the CPU oracle, bit for bit.
Usage: python scripts/fuzz_sharpness.py [batches] [seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

from frhot import synth  # noqa: E402
from frhot.engine import Context  # noqa: E402
from oracle import quality as oq  # noqa: E402


def rand_dim(rng):
    kind = rng.integers(0, 8)
    if kind == 0:
        return int(rng.integers(1, 9))
    if kind == 1:
        return int(rng.choice([15, 16, 17, 31, 32, 33, 255, 256, 257, 258, 259, 260, 511, 512, 513, 1023, 1025]))
    if kind == 2:
        return int(rng.integers(600, 2000))
    return int(rng.integers(8, 600))


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = Context(device=0, max_batch=12)
    n = 0
    for b in range(batches):
        imgs = []
        for _ in range(int(rng.integers(1, 13))):
            h, w = rand_dim(rng), rand_dim(rng)
            style = rng.integers(0, 4)
            if style == 0:
                im = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
            elif style == 1:
                im = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)
            else:
                im = synth.crop(rng, h, w)
            view = rng.integers(0, 4)
            if view == 1 and h >= 4:
                im = im[int(rng.integers(0, 2))::2]                       # every other row
            elif view == 2 and w >= 6:
                lo = int(rng.integers(0, w // 2))
                im = im[:, lo:lo + int(rng.integers(1, w - lo + 1))]       # a column band of a wider image
            imgs.append(im)
        got = ctx.crop_sharpness_sums(imgs)
        for i, im in enumerate(imgs):
            want = oq.laplacian_sums(np.ascontiguousarray(im))
            assert tuple(int(v) for v in got[i]) == want, f"batch {b} crop {i} {im.shape} strides {im.strides}: {got[i]} != {want}"
        n += len(imgs)
    ctx.close()
    print(f"fuzz_sharpness: {n} crops bit-exact")


if __name__ == "__main__":
    main()
