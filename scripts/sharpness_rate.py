"""Rate of the blur-gate kernel (crop_sharpness_kernel) on device-resident crops.

  python scripts/sharpness_rate.py [B] [lo] [hi]

Prints microseconds per launch (HIP events on the context's stream, memset included) and the
crop bytes per second (h * w * 3 bytes per crop are the kernel's algorithmic traffic)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "face-recognition_amd"))
from frhot import synth                      # noqa: E402
from frhot._lib import check, load           # noqa: E402
from frhot.engine import Context             # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    lo = int(sys.argv[2]) if len(sys.argv) > 2 else 112
    hi = int(sys.argv[3]) if len(sys.argv) > 3 else 448
    stream = torch.cuda.Stream()
    ctx = Context(device=0, max_batch=B, stream=stream.cuda_stream)
    imgs = synth.crops(seed=3, n=B, lo=lo, hi=hi)
    packed, offsets, hw, strides = synth.pack_crops(imgs)
    dev = torch.from_numpy(packed).cuda()
    out = torch.zeros((B, 2), dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    lib = load()
    args = (ctx.handle, C.c_void_p(dev.data_ptr()), offsets.ctypes.data_as(C.POINTER(C.c_int64)),
            hw.ctypes.data_as(C.POINTER(C.c_int32)), strides.ctypes.data_as(C.POINTER(C.c_int32)), B,
            C.c_void_p(out.data_ptr()))
    nbytes = int((hw[:, 0].astype(np.int64) * hw[:, 1] * 3).sum())
    with torch.cuda.stream(stream):
        for _ in range(5):
            check(lib.frhot_crop_sharpness_dev(*args))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record(stream)
        for _ in range(n):
            check(lib.frhot_crop_sharpness_dev(*args))
        e1.record(stream)
    e1.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    want = ctx.crop_sharpness_sums(imgs)
    assert np.array_equal(out.cpu().numpy(), want)
    print(f"B={B} sides {lo}..{hi}: {nbytes / 1e6:.1f} MB of crops, {us:.1f} us per launch, "
          f"{nbytes / us / 1e6:.2f} TB/s")


if __name__ == "__main__":
    main()
