import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "face-recognition_amd"))
import numpy as np
from frhot import synth
from frhot.engine import Context
from frhot.weights import synthetic_weights, quantize_fp16, pack_blob
from oracle import preprocess as op
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_hotpath import _oracle_layers

w = synthetic_weights(0xFACE)
wq = quantize_fp16(w)
ctx = Context(0, 16)
ctx.load_model(pack_blob(w), 1/16)
imgs = synth.crops(seed=3, n=2, lo=150, hi=260)
ctx.embed(imgs)
x = np.stack([op.preprocess(im) for im in imgs])
want = _oracle_layers(x, wq)
for l in range(13):
    got = ctx.debug_read(10 + l, 2).astype(np.float32) * 16
    if l < 12: got = got[:, 1:-1, 1:-1]
    ref = want[l]
    bad = ~np.isfinite(got)
    err = np.abs(np.where(bad, 0, got) - ref)
    print("layer", l, got.shape, "nonfinite", bad.sum(), "maxerr", err.max(), "refmax", np.abs(ref).max(),
          "mean rel", err.mean() / np.abs(ref).mean())
    if bad.any():
        idx = np.argwhere(bad)
        print("  first bad", idx[:5], "chan hist", np.bincount(idx[:, 3], minlength=ref.shape[3])[:16])
    if err.max() > 0.01 * np.abs(ref).max():
        idx = np.argwhere(err > 0.01 * np.abs(ref).max())
        print("  n wrong", len(idx), "first", idx[:8].tolist())
        i = tuple(idx[0]); print("  got", got[i], "ref", ref[i])
        print("  wrong by chan%16", np.bincount(idx[:, 3] % 16, minlength=16), "by x%16", np.bincount(idx[:, 2] % 16, minlength=16))
