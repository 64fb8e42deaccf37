#!/usr/bin/env python3
"""Developer tool: randomised bit-exactness sweep of the preprocessing kernels against the CPU oracle.

Random crop geometries (1 pixel to ~1500 px a side, extreme aspect ratios, near-1x and near-2x
scales, row-strided views), gamma on or off, channel flip on or off; the resized bytes and the fp16
network input must equal the oracle's bit for bit.
Usage: python scripts/fuzz_preprocess.py [batches] [seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

from frhot.engine import Context  # noqa: E402
from frhot.weights import pack_blob, synthetic_weights  # noqa: E402
from oracle import preprocess as op  # noqa: E402


def rand_dim(rng):
    kind = rng.integers(0, 8)
    if kind == 0:
        return int(rng.integers(1, 8))
    if kind == 1:
        return int(rng.choice([223, 224, 225, 447, 448, 449, 112, 896]))
    if kind == 2:
        return int(rng.integers(600, 1500))
    return int(rng.integers(8, 600))


def main():
    batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    ctx = Context(device=0, max_batch=8)
    ctx.load_model(pack_blob(synthetic_weights(seed=0xFACE)), input_scale=1.0 / 16.0)
    n = 0
    for b in range(batches):
        gamma, flip = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        imgs = []
        for _ in range(8):
            h, w = rand_dim(rng), rand_dim(rng)
            while min(op.fit_size(h, w)) <= 0:                  # resizes to nothing: refused, like cv2 refuses it
                h, w = rand_dim(rng), rand_dim(rng)
            style = rng.integers(0, 4)
            if style == 0:
                im = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
            elif style == 1:
                im = np.full((h, w, 3), int(rng.integers(0, 256)), np.uint8)
            elif style == 2:                                    # smooth gradient
                yy, xx = np.mgrid[0:h, 0:w]
                im = np.stack([(yy * 3 + xx) % 256, (xx * 2) % 256, (yy + xx * 5) % 256], -1).astype(np.uint8)
            else:                                               # a view into a larger picture
                bigp = rng.integers(0, 256, size=(h + 7, w + 9, 3), dtype=np.uint8)
                im = bigp[3:3 + h, 4:4 + w]
            imgs.append(im)
        ctx.embed(imgs, gamma=gamma, flip=flip)
        u8 = ctx.debug_read(0, len(imgs))
        x0 = ctx.debug_read(1, len(imgs))
        for i, im in enumerate(imgs):
            imc = np.ascontiguousarray(im)
            want_u8 = op.preprocess_u8(imc, apply_gamma=gamma, flip=flip)
            assert np.array_equal(u8[i], want_u8), f"batch {b} crop {i} {im.shape} gamma={gamma} flip={flip}: bytes differ"
            want = (op.preprocess(imc, apply_gamma=gamma, flip=flip) * np.float32(1.0 / 16.0)).astype(np.float16)
            assert np.array_equal(x0[i, 1:225, 1:225, :3], want), f"batch {b} crop {i} {im.shape}: network input differs"
            n += 1
        print(f"batch {b}: 8 crops ok (gamma={gamma}, flip={flip})", flush=True)
    ctx.close()
    print(f"fuzz_preprocess: {n} crops bit-exact")


if __name__ == "__main__":
    main()
