"""Oracle: crop -> gamma LUT -> DeepFace VGG-Face preprocessing (CPU, numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for the
third-party steps: no fixture of the reference covers them.

Reference call sites this file follows:
  * crop slice and gamma order ........ app/src/services/face_service.py:145-150,178-181,221-223
  * gamma LUT ......................... app/src/utils/legacy.py:106-110
  * DeepFace.represent parameters ..... app/src/utils/legacy.py:52-59
Third-party behaviour restated (deepface 0.0.93 modules/representation.py and
modules/preprocessing.py; opencv-python 4.10.0.84 imgproc resize):
  * channel reverse before resize, aspect-fit resize with truncated dsize,
    centred zero pad, conditional /255, "VGGFace2" mean subtraction;
  * cv2.resize INTER_LINEAR on uint8: 11-bit fixed-point coefficients,
    half-pixel centres, edge clamp; exact 2x downscale -> 2x2 area average.
"""
from __future__ import annotations

import numpy as np

TARGET = 224
VGGFACE2_MEAN = (91.4953, 103.8827, 131.0912)
COEF_BITS = 11
COEF_ONE = 1 << COEF_BITS  # 2048


def crop_box(img: np.ndarray, box) -> np.ndarray:
    """face_service.py:147-150,180-181: x,y,w,h -> ints, python slice, uint8."""
    x, y, w, h = box
    x1, y1, x2, y2 = int(x), int(y), int(x + w), int(y + h)
    return img[y1:y2, x1:x2].astype("uint8")


def gamma_table(gamma: float = 1.5) -> np.ndarray:
    """legacy.py:108-109: float64 power, truncating cast to uint8."""
    inv = 1.0 / gamma
    return np.array([(i / 255.0) ** inv * 255 for i in np.arange(0, 256)]).astype("uint8")


def adjust_gamma(image: np.ndarray, gamma: float = 1.0) -> np.ndarray:
    """legacy.py:106-110 (cv2.LUT == table lookup per byte)."""
    return gamma_table(gamma)[image]


def _axis_coeffs(dst: int, src: int, clamp_frac: bool):
    """Source index and 11-bit weights for one axis of cv2.resize INTER_LINEAR.

    Horizontal taps (clamp_frac=True) zero the fraction when the left tap falls
    outside [0, src-1); vertical taps keep the fraction and clamp the row
    indices instead.  Either way both taps land on the same edge pixel.
    """
    inv_scale = float(dst) / float(src)
    scale = 1.0 / inv_scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_frac:
        lo = s < 0
        f[lo] = 0.0
        s[lo] = 0
        hi = s >= src - 1
        f[hi] = 0.0
        s[hi] = src - 1
    w0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_ONE)).astype(np.int32)
    w1 = np.rint(f * np.float32(COEF_ONE)).astype(np.int32)
    i0 = np.clip(s, 0, src - 1)
    i1 = np.clip(s + 1, 0, src - 1)
    return i0, i1, w0, w1


def resize_linear_u8(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    """cv2.resize(src, (dw, dh)) for uint8 HxWxC with the default INTER_LINEAR."""
    assert src.dtype == np.uint8 and src.ndim == 3
    sh, sw = src.shape[:2]
    if dw <= 0 or dh <= 0:
        raise ValueError("resize target has a zero dimension")
    if sw == 2 * dw and sh == 2 * dh:
        # INTER_LINEAR with both scale factors exactly 2 takes the area path.
        s = src.astype(np.int32)
        acc = s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2]
        return ((acc + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _axis_coeffs(dw, sw, clamp_frac=True)
    y0, y1, b0, b1 = _axis_coeffs(dh, sh, clamp_frac=False)
    s = src.astype(np.int32)
    rows = s[:, x0, :] * a0[None, :, None] + s[:, x1, :] * a1[None, :, None]
    r0 = rows[y0] >> 4
    r1 = rows[y1] >> 4
    out = (((b0[:, None, None] * r0) >> 16) + ((b1[:, None, None] * r1) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def fit_size(h: int, w: int, target: int = TARGET):
    """deepface resize_image: factor in float64, dsize truncated."""
    f0 = target / h
    f1 = target / w
    f = min(f0, f1)
    return int(w * f), int(h * f)  # (dw, dh)


def resize_pad_u8(img: np.ndarray, target: int = TARGET) -> np.ndarray:
    """Aspect-fit resize and centred zero pad to target x target (uint8)."""
    h, w = img.shape[:2]
    dw, dh = fit_size(h, w, target)
    r = resize_linear_u8(img, dw, dh)
    d0 = target - dh
    d1 = target - dw
    out = np.zeros((target, target, img.shape[2]), dtype=np.uint8)
    out[d0 // 2:d0 // 2 + dh, d1 // 2:d1 // 2 + dw] = r
    return out


def normalize_vggface2(u8: np.ndarray) -> np.ndarray:
    """float32 conversion, conditional /255, *255, per-channel mean subtraction."""
    x = u8.astype(np.float32)
    if x.max() > 1:
        x = (x / np.float32(255.0)).astype(np.float32)
    x = x * np.float32(255.0)
    for c in range(3):
        x[..., c] -= np.float32(VGGFACE2_MEAN[c])
    return x


def preprocess(crop_bgr: np.ndarray, apply_gamma: bool = False, flip: bool = True) -> np.ndarray:
    """uint8 HxWx3 crop -> float32 224x224x3 network input (NHWC, one image)."""
    img = adjust_gamma(crop_bgr, 1.5) if apply_gamma else crop_bgr
    if flip:
        img = img[:, :, ::-1]
    return normalize_vggface2(resize_pad_u8(np.ascontiguousarray(img)))


def preprocess_u8(crop_bgr: np.ndarray, apply_gamma: bool = False, flip: bool = True) -> np.ndarray:
    """The uint8 224x224x3 image right before float conversion (exact stage)."""
    img = adjust_gamma(crop_bgr, 1.5) if apply_gamma else crop_bgr
    if flip:
        img = img[:, :, ::-1]
    return resize_pad_u8(np.ascontiguousarray(img))
