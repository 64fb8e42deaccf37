"""Random frames through the GPU detector against the oracle: letterbox bit for bit, decoded
candidates (suppression off) and regions.  Usage: fuzz_detector.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "face-recognition_amd"), os.path.join(ROOT, "tests")]


def main():
    from det_util import calibrated_weights
    from frhot import detector as D
    from frhot.engine import Context
    from oracle import yolo_face as O
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    w, wq = calibrated_weights(seed=3, cls_bias=-6.0)
    ctx = Context(device=0, max_batch=1)
    det = D.Detector(ctx, D.pack_blob(w), max_batch=4)
    bad = 0
    for case in range(cases):
        kind = rng.integers(0, 5)
        if kind == 0:
            h, w_ = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        elif kind == 1:
            h, w_ = int(rng.integers(600, 1400)), int(rng.integers(600, 2000))
        elif kind == 2:
            h, w_ = int(rng.integers(8, 64)), int(rng.integers(400, 1500))
        else:
            h, w_ = int(rng.integers(100, 720)), int(rng.integers(100, 720))
        base = rng.random((max(2, h // 16), max(2, w_ // 16), 3))
        im = np.kron(base, np.ones((16, 16, 1)))[:h, :w_]
        im = np.ascontiguousarray((im * 255).astype(np.uint8))
        n = int(rng.integers(1, 4))
        frames = [im] + [np.ascontiguousarray(im[::-1]) for _ in range(n - 1)]
        want_x = O.letterbox(im)[0]
        got = det.detect(frames, conf=0.25, iou=1.0, max_faces=300)
        canvas = det.debug_read("in", n, want_x.shape[0], want_x.shape[1], 4)
        ok = np.array_equal(canvas[0, :, :, :3], want_x.astype(np.float16))
        boxes, scores, kpts = O.decode(O.forward(want_x[None], wq))[0]
        far = np.abs(scores - 0.25) > 0.03
        n_ref = int(far.sum())
        rows = got[0]
        # every clear reference candidate has a device row with its score (to 0.02); counts agree up to the borderline ones
        hit = 0
        for s in scores[far][:300]:
            hit += bool(len(rows)) and bool((np.abs(rows[:, 4] - s) <= 0.02).any())
        ok = ok and (len(scores) > 300 or hit >= n_ref - 1) and abs(len(rows) - min(len(scores), 300)) <= max(2, len(scores) // 20)
        ref_regions = O.regions(im, wq)
        got_regions = det.regions(im)
        ok = ok and abs(len(ref_regions) - len(got_regions)) <= max(1, len(ref_regions) // 10)
        print(f"case {case}: {h}x{w_} x{n} canvas {want_x.shape[:2]} candidates {len(scores)} regions {len(ref_regions)}/{len(got_regions)}"
              f" {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += not ok
    det.close()
    ctx.close()
    if bad:
        raise SystemExit(f"fuzz_detector: {bad} of {cases} cases differ")
    print(f"fuzz_detector: {cases} cases passed")


main()
