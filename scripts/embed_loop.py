"""Developer tool: frhot_embed_dev on device-resident crops in a loop, nothing else on the GPU:
   python scripts/embed_loop.py [batch] [steps] [gap] [data]     gap=1 waits for each batch before the next is enqueued;
   data=noise: 224 x 224 crops of uniform random bytes, uniform weights with the He deviation, zero biases (what scripts/kbench feeds the kernels)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'face-recognition_amd'))
import numpy as np
import torch
from frhot import synth
from frhot.engine import Context
from frhot.weights import pack_blob, synthetic_weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
gap = len(sys.argv) > 3 and sys.argv[3] == '1'
noise = len(sys.argv) > 4 and sys.argv[4] == 'noise'
dev = torch.device('cuda:0')
stream = torch.cuda.Stream(device=dev)
with torch.cuda.stream(stream):
    ctx = Context(device=0, max_batch=B, stream=stream.cuda_stream)
    weights = synthetic_weights(seed=1)
    images = synth.crops(seed=4, n=B, lo=112, hi=448)
    if noise:
        rng = np.random.default_rng(5)
        for k in weights:
            if k.endswith('.b'):
                weights[k][:] = 0
            else:
                sd = float(weights[k].std())
                weights[k] = rng.uniform(-sd * 3 ** 0.5, sd * 3 ** 0.5, size=weights[k].shape).astype(np.float32)
        images = [rng.integers(0, 256, size=(224, 224, 3), dtype=np.uint8) for _ in range(B)]
    ctx.load_model(pack_blob(weights), input_scale=1 / 16)
    buf, offsets, hw, strides = synth.pack_crops(images)
    crops_dev = torch.from_numpy(buf).to(dev)
    q = torch.empty((B, 512), dtype=torch.float16, device=dev)
    flags = Context.flags(gamma=True, flip=True)
    for _ in range(3):
        ctx.embed_dev(crops_dev.data_ptr(), offsets, hw, strides, flags, None, q.data_ptr(), 512)
    torch.cuda.synchronize()
    ctx.timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.embed_dev(crops_dev.data_ptr(), offsets, hw, strides, flags, None, q.data_ptr(), 512)
        if gap:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    st = ctx.stage_ms()
    import hashlib
    digest = hashlib.sha256(q.cpu().numpy().tobytes()).hexdigest()[:12]
    print('batch %d, %d steps%s: %.3f ms per batch; stages %s; descriptors %s' % (B, steps, ', each awaited' if gap else '', dt * 1e3,
                                                                               {k: round(v, 3) for k, v in st.items()}, digest))
    ctx.close()
