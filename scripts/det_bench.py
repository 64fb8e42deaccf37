"""Throughput of the face detector: frames of 480 x 640 through Detector.detect (host frames in,
regions out), and the per-kernel picture under rocprofv3 if run beneath it."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition_amd")]


def main():
    from frhot import detector as D, synth as S
    from frhot.engine import Context
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    ctx = Context(device=0, max_batch=1)
    det = D.Detector(ctx, D.pack_blob(D.synthetic_weights(3)), max_batch=batch)
    frames = [np.ascontiguousarray(f[:480, :640]) for f in S.crops(seed=1, n=batch, lo=640, hi=700)]
    for _ in range(3):
        det.detect(frames)
    t = time.perf_counter()
    for _ in range(iters):
        det.detect(frames)
    dt = (time.perf_counter() - t) / iters
    fl = 2.0 * D.macs_for(480, 640) * batch
    print(f"detect: batch {batch}: {dt * 1e3:.3f} ms per call, {batch / dt:.0f} frames/s, {fl / dt / 1e12:.1f} TFLOP/s "
          f"({fl / dt / 2.5e15:.3f} of 2.5 PFLOP/s)")
    det.close()
    ctx.close()


main()
