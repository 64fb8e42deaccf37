"""Shared by the detector tests: random weights of the architecture, calibrated by the oracle so
that activations keep unit scale through the network (test infrastructure: uses the oracle)."""
import numpy as np


def calibrated_weights(seed: int = 3, cls_bias: float = -6.0):
    """(float32 weights, the same with fp16-rounded kernels as the library stores them)."""
    from frhot import detector as D, synth as S
    from oracle import yolo_face as O
    w = D.synthetic_weights(seed, cls_bias=cls_bias)
    imgs = S.crops(seed=seed + 100, n=2, lo=400, hi=640)
    x = np.stack([O.letterbox(im, auto=False)[0] for im in imgs])
    with np.errstate(over="ignore"):
        O.forward(x, w, calibrate=True)
    wq = {k: (v.astype(np.float16).astype(np.float32) if k.endswith(".w") else v) for k, v in w.items()}
    return w, wq


def frames(seed: int, n: int, lo: int = 240, hi: int = 900):
    from frhot import synth as S
    return S.crops(seed=seed, n=n, lo=lo, hi=hi)
