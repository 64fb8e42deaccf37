"""Developer tool: frhot_embed at small batch sizes: the host call as served, and its stages."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/face-recognition_amd')
import numpy as np
from frhot import synth
from frhot.engine import Context
from frhot.weights import pack_blob, synthetic_weights
ctx = Context(device=0, max_batch=64)
ctx.load_model(pack_blob(synthetic_weights(seed=1)), input_scale=1/16)
for b in (1, 4, 8, 16, 32, 64):
    imgs = synth.crops(seed=3, n=b, lo=150, hi=300)
    for _ in range(3): ctx.embed(imgs)
    t0 = time.perf_counter()
    for _ in range(50): ctx.embed(imgs)
    dt = (time.perf_counter() - t0) / 50
    ctx.timing(True)
    for _ in range(20): ctx.embed(imgs)
    st = ctx.stage_ms(); ctx.timing(False)
    print(b, 'host call ms %.3f' % (dt * 1e3), {k: round(v, 3) for k, v in st.items()})
ctx.close()
