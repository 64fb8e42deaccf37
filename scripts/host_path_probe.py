import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "face-recognition_amd"))
import numpy as np
from frhot import synth
from frhot.engine import Context, Gallery, _crop_args
from frhot.weights import pack_blob, synthetic_weights
B=256
ctx = Context(device=0, max_batch=B)
ctx.load_model(pack_blob(synthetic_weights(seed=0xFACE)), input_scale=1.0 / 16.0)
gal = Gallery(ctx, 512, 1_000_000)
gal.fill_synthetic(seed=1, row0=0, n=1_000_000, nkeys=0, dup_every=100)
images = synth.crops(seed=0xFACE0003, n=B, lo=112, hi=448)
def t(f, n=5):
    f(); t0=time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter()-t0)/n*1e3
print("crop_args ms", t(lambda: _crop_args(images)))
print("recognise ms", t(lambda: ctx.recognise(gal, images, None, 5, gamma=True)))
print("recognise+emb ms", t(lambda: ctx.recognise(gal, images, None, 5, gamma=True, want_embeddings=True)))
print("embed ms", t(lambda: ctx.embed(images, gamma=True)))
ctx.timing(True)
ctx.recognise(gal, images, None, 5, gamma=True); ctx.sync()
print(ctx.stage_ms())
big = np.concatenate([im.reshape(-1) for im in images])
t0=time.perf_counter(); c = big.copy(); print("numpy copy of", big.nbytes/1e6, "MB ms", (time.perf_counter()-t0)*1e3)
print("cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
