"""Oracle: VGG-Face descriptor (VGG-16 to post-ReLU fc7) + L2 norm, CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the
architecture lives in deepface 0.0.93 (models/facial_recognition/VGGFace.py),
which is not installed here, and the reference holds no embedding fixture.

Reference call site: app/src/utils/legacy.py:52-59 (model_name="VGG-Face");
the descriptor length 4096 is pinned by qdrant_database_FE/app.py:116,258,312.

Restated architecture: 13 x (3x3 conv, pad 1, bias, ReLU) with 2x2/2 max-pool
after conv 2, 4, 7, 10, 13; fc6 = 7x7 valid conv 512->4096 + ReLU; fc7 = 1x1
conv 4096->4096 + ReLU; descriptor = flattened fc7 output; then
x / ||x|| in float64 (unchanged when the norm is zero).

Weight layout used by this repo ("KRSC"): conv weight [Cout, 3, 3, Cin],
fc6 [4096, 7, 7, 512], fc7 [4096, 4096]; biases [Cout]; all float32 here.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

# (name, Cin, Cout, pool_after)
CONV_LAYERS = [
    ("conv1_1", 3, 64, False), ("conv1_2", 64, 64, True),
    ("conv2_1", 64, 128, False), ("conv2_2", 128, 128, True),
    ("conv3_1", 128, 256, False), ("conv3_2", 256, 256, False), ("conv3_3", 256, 256, True),
    ("conv4_1", 256, 512, False), ("conv4_2", 512, 512, False), ("conv4_3", 512, 512, True),
    ("conv5_1", 512, 512, False), ("conv5_2", 512, 512, False), ("conv5_3", 512, 512, True),
]
EMBED_DIM = 4096
MACS_PER_FACE = 15_466_168_320


def forward(x_nhwc: np.ndarray, weights: dict, threads: int | None = None) -> np.ndarray:
    """float32 [B,224,224,3] -> float32 [B,4096] (post-ReLU fc7, not normalised)."""
    if threads is not None:
        torch.set_num_threads(threads)
    x = torch.from_numpy(np.ascontiguousarray(x_nhwc, dtype=np.float32)).permute(0, 3, 1, 2)
    with torch.no_grad():
        for name, _cin, _cout, pool in CONV_LAYERS:
            w = torch.from_numpy(np.asarray(weights[name + ".w"], dtype=np.float32)).permute(0, 3, 1, 2)
            b = torch.from_numpy(np.asarray(weights[name + ".b"], dtype=np.float32))
            x = F.relu(F.conv2d(x, w, b, padding=1))
            if pool:
                x = F.max_pool2d(x, 2, 2)
        w6 = torch.from_numpy(np.asarray(weights["fc6.w"], dtype=np.float32)).permute(0, 3, 1, 2)
        x = F.relu(F.conv2d(x, w6, torch.from_numpy(np.asarray(weights["fc6.b"], dtype=np.float32))))
        w7 = torch.from_numpy(np.asarray(weights["fc7.w"], dtype=np.float32))[:, :, None, None]
        x = F.relu(F.conv2d(x, w7, torch.from_numpy(np.asarray(weights["fc7.b"], dtype=np.float32))))
        return x.flatten(1).numpy()


def l2_normalize(x: np.ndarray) -> np.ndarray:
    """deepface verification.l2_normalize on the float64 list; rows independently."""
    x = np.asarray(x, dtype=np.float64)
    if x.ndim == 1:
        n = np.linalg.norm(x)
        return x if n == 0 else x / n
    n = np.linalg.norm(x, axis=1, keepdims=True)
    n[n == 0] = 1.0
    return x / n


def embed(x_nhwc: np.ndarray, weights: dict, threads: int | None = None) -> np.ndarray:
    """Unit-norm float64 descriptors [B,4096] for preprocessed inputs."""
    return l2_normalize(forward(x_nhwc, weights, threads))
