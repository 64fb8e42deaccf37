"""VGG-Face weight containers: the flat blob libfrhot.so loads, synthetic weights,
and a converter from the Keras file DeepFace downloads.

Blob layout (little endian): 8-byte magic b"FRHOTW01", u32 version (1), u32 count,
`count` directory entries {char name[16]; u32 dtype (0 fp16, 1 fp32); u32 ndim;
u32 dims[4]; u64 offset; u64 nbytes}, then the tensors at 64-byte aligned offsets.
Tensors: "<layer>.w" fp16 in [Cout, kh, kw, Cin] order (fc7: [4096, 4096]) and
"<layer>.b" fp32 for conv1_1 .. conv5_3, fc6, fc7.

The reference obtains the network through DeepFace.represent(model_name="VGG-Face")
(app/src/utils/legacy.py:52-54); its weight file is vgg_face_weights.h5
(app/Dockerfile:79-81).
"""
from __future__ import annotations

import struct

import numpy as np

# (name, kh, kw, Cin, Cout)
LAYERS = [
    ("conv1_1", 3, 3, 3, 64), ("conv1_2", 3, 3, 64, 64),
    ("conv2_1", 3, 3, 64, 128), ("conv2_2", 3, 3, 128, 128),
    ("conv3_1", 3, 3, 128, 256), ("conv3_2", 3, 3, 256, 256), ("conv3_3", 3, 3, 256, 256),
    ("conv4_1", 3, 3, 256, 512), ("conv4_2", 3, 3, 512, 512), ("conv4_3", 3, 3, 512, 512),
    ("conv5_1", 3, 3, 512, 512), ("conv5_2", 3, 3, 512, 512), ("conv5_3", 3, 3, 512, 512),
    ("fc6", 7, 7, 512, 4096), ("fc7", 1, 1, 4096, 4096),
]
MAGIC = b"FRHOTW01"
_ENTRY = struct.Struct("<16sII4IQQ")
assert _ENTRY.size == 56


def weight_shape(name, kh, kw, cin, cout):
    return (cout, cin) if name == "fc7" else (cout, kh, kw, cin)


def synthetic_weights(seed: int = 0xFACE, bias_scale: float = 0.5) -> dict:
    """Random-init weights of the VGG-Face architecture (float32, KRSC).

    He-normal weights keep the activation scale of the mean-subtracted pixels
    through the stack; biases are small normal values.
    """
    rng = np.random.default_rng(seed)
    out = {}
    for name, kh, kw, cin, cout in LAYERS:
        fan_in = kh * kw * cin
        std = np.float32(np.sqrt(2.0 / fan_in))
        w = rng.standard_normal(size=weight_shape(name, kh, kw, cin, cout), dtype=np.float32) * std
        out[name + ".w"] = w
        out[name + ".b"] = (rng.standard_normal(size=(cout,), dtype=np.float32) * np.float32(bias_scale))
    return out


def scaled_weights(weights: dict, alpha: float) -> dict:
    """The same network with every activation multiplied by `alpha` (> 0): conv1_1's kernel and
    every bias are scaled, which by the positive homogeneity of ReLU(Wx + b) scales every layer's
    output and leaves the unit descriptor unchanged.  Drives the activation range of a test."""
    a = np.float32(alpha)
    out = {k: v.copy() for k, v in weights.items()}
    out["conv1_1.w"] = out["conv1_1.w"] * a
    for name, *_ in LAYERS:
        out[name + ".b"] = out[name + ".b"] * a
    return out


def growing_weights(seed: int = 0xFACE, gain: float = 1.25, bias_scale: float = 0.5) -> dict:
    """He-normal kernels whose standard deviation grows by `gain` per convolution layer, so that
    activations rise through the stack and peak in conv5_x / fc6 instead of staying level."""
    out = synthetic_weights(seed, bias_scale)
    g = np.float32(1.0)
    for name, *_ in LAYERS[:13]:
        g = g * np.float32(gain)
        out[name + ".w"] = out[name + ".w"] * np.float32(gain)
        out[name + ".b"] = out[name + ".b"] * g
    for name in ("fc6", "fc7"):
        out[name + ".b"] = out[name + ".b"] * g
    return out


def heavy_tailed_weights(seed: int = 0xFACE, df: float = 3.0, bias_scale: float = 0.5) -> dict:
    """Kernels drawn from Student's t (df degrees of freedom) scaled to the He variance: a few
    taps per filter are many standard deviations large, as trained filters' are."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, kh, kw, cin, cout in LAYERS:
        fan_in = kh * kw * cin
        std = np.sqrt(2.0 / fan_in) / np.sqrt(df / (df - 2.0))
        w = rng.standard_t(df, size=weight_shape(name, kh, kw, cin, cout)).astype(np.float32) * np.float32(std)
        out[name + ".w"] = w
        out[name + ".b"] = rng.standard_normal(size=(cout,), dtype=np.float32) * np.float32(bias_scale)
    return out


def discriminative_weights(weights: dict, fc6_sample: np.ndarray, thr_mult: float = 1.6) -> dict:
    """The same stack with an fc7 that tells inputs apart.  Descriptors of a randomly initialised
    ReLU network all point the same way (pairwise cosine 0.94-0.98): every search hit passes the
    0.54 threshold.  Here fc7 ignores the direction the fc6 outputs share (the mean of
    `fc6_sample`, [n, 4096] fc6 activations of a few inputs, is projected out of its kernel) and
    its bias is -thr_mult standard deviations of the remaining response, so that a few percent
    of the units fire and which ones depends on the input: different inputs score well below
    0.54, an input scores 1 against itself."""
    h = np.asarray(fc6_sample, dtype=np.float64)
    m = h.mean(axis=0)
    w7 = np.asarray(weights["fc7.w"], dtype=np.float64)
    w7p = w7 - np.outer(w7 @ m, m) / float(m @ m)
    z = h @ w7p.T
    out = {k: v.copy() for k, v in weights.items()}
    out["fc7.w"] = w7p.astype(np.float32)
    out["fc7.b"] = np.full(w7.shape[0], -thr_mult * z.std(), dtype=np.float32)
    return out


def quantize_fp16(weights: dict) -> dict:
    """Weights as the library stores them: fp16-rounded matrices, fp32 biases (returned as float32)."""
    return {k: (v.astype(np.float16).astype(np.float32) if k.endswith(".w") else v.astype(np.float32))
            for k, v in weights.items()}


def pack_blob(weights: dict) -> np.ndarray:
    """dict of float arrays -> uint8 blob for frhot_model_load."""
    entries = []
    for name, kh, kw, cin, cout in LAYERS:
        w = np.ascontiguousarray(weights[name + ".w"]).astype(np.float16)
        if w.shape != weight_shape(name, kh, kw, cin, cout):
            raise ValueError(f"{name}.w has shape {w.shape}")
        b = np.ascontiguousarray(weights[name + ".b"]).astype(np.float32)
        if b.shape != (cout,):
            raise ValueError(f"{name}.b has shape {b.shape}")
        entries.append((name + ".w", 0, w))
        entries.append((name + ".b", 1, b))
    return pack_entries(entries)


def pack_entries(entries) -> np.ndarray:
    """[(name (at most 15 characters), dtype code (0 fp16, 1 fp32), array)] -> uint8 blob."""
    head = 16 + _ENTRY.size * len(entries)
    off = (head + 63) & ~63
    layout = []
    for nm, dt, arr in entries:
        layout.append((nm, dt, arr, off))
        off = (off + arr.nbytes + 63) & ~63
    blob = np.zeros(off, dtype=np.uint8)
    blob[:8] = np.frombuffer(MAGIC, dtype=np.uint8)
    blob[8:16] = np.frombuffer(struct.pack("<II", 1, len(entries)), dtype=np.uint8)
    pos = 16
    for nm, dt, arr, o in layout:
        dims = list(arr.shape) + [0] * (4 - arr.ndim)
        rec = _ENTRY.pack(nm.encode("ascii"), dt, arr.ndim, *dims, o, arr.nbytes)
        blob[pos:pos + _ENTRY.size] = np.frombuffer(rec, dtype=np.uint8)
        pos += _ENTRY.size
        blob[o:o + arr.nbytes] = arr.reshape(-1).view(np.uint8)
    return blob


def from_keras_h5(path: str) -> dict:
    """vgg_face_weights.h5 (Keras, kernels [kh, kw, Cin, Cout]) -> KRSC float32 dict.

    Needs h5py, which this image does not ship; the function exists so that a
    host holding the real file can produce a blob with pack_blob().  The Keras
    model names its convolution layers in definition order; the first 13 are the
    3x3 stack, then fc6 (7x7), fc7 (1x1) and the discarded 2622-way classifier.
    """
    import h5py  # noqa: F401  (absent here by design: raises ImportError)

    out = {}
    with h5py.File(path, "r") as f:
        names = [n.decode() if isinstance(n, bytes) else n for n in f.attrs["layer_names"]]
        convs = []
        for n in names:
            g = f[n]
            wn = [x.decode() if isinstance(x, bytes) else x for x in g.attrs.get("weight_names", [])]
            if len(wn) == 2:
                convs.append((np.asarray(g[wn[0]]), np.asarray(g[wn[1]])))
        if len(convs) < 15:
            raise ValueError("expected at least 15 weighted layers in the Keras file")
        for (name, kh, kw, cin, cout), (k, b) in zip(LAYERS, convs[:15]):
            if k.shape != (kh, kw, cin, cout):
                raise ValueError(f"{name}: Keras kernel shape {k.shape}")
            w = np.transpose(k, (3, 0, 1, 2)).astype(np.float32)
            out[name + ".w"] = w.reshape(cout, cin) if name == "fc7" else w
            out[name + ".b"] = b.astype(np.float32)
    return out
