#!/usr/bin/env python3
"""Developer tool: every batch size gives every crop the descriptor it gets in a batch of 16, bit for bit
(partial last tiles of every conv layer, both tile walk directions, split-K FC with few rows), through the host call and through
the device-pointer call (whose large batches launch the early layers with a share of the CUs per half)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from frhot import synth  # noqa: E402
from frhot.engine import Context  # noqa: E402
from frhot.weights import pack_blob, synthetic_weights  # noqa: E402


def main():
    pool = synth.crops(seed=99, n=64, lo=112, hi=200)
    blob = pack_blob(synthetic_weights(seed=0xFACE))
    small = Context(device=0, max_batch=16)
    small.load_model(blob, input_scale=1.0 / 16.0)
    want = small.embed(pool, gamma=True)
    small.close()
    big = Context(device=0, max_batch=300)
    big.load_model(blob, input_scale=1.0 / 16.0)
    rng = np.random.default_rng(3)
    sizes = [1, 2, 3, 4, 5, 6, 7, 9, 13, 16, 17, 31, 33, 63, 64, 65, 100, 127, 129, 199, 255, 257, 300]
    for b in sizes:
        idx = rng.integers(0, 64, size=b)
        got = big.embed([pool[i] for i in idx], gamma=True)
        assert np.array_equal(got, want[idx]), f"batch {b}: descriptors differ"
        buf, offsets, hw, strides = synth.pack_crops([pool[i] for i in idx])
        crops_dev = torch.from_numpy(buf).to("cuda:0")
        out = torch.empty((b, 4096), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        big.embed_dev(crops_dev.data_ptr(), offsets, hw, strides, Context.flags(gamma=True, flip=True), out.data_ptr(), None)
        big.sync()
        assert np.array_equal(out.cpu().numpy(), want[idx]), f"batch {b}: descriptors of the device-pointer call differ"
        print(f"batch {b}: ok", flush=True)
    big.close()
    print("fuzz_batches: all batch sizes bit-identical")


if __name__ == "__main__":
    main()
