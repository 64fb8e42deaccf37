"""The face detector of the reference's DeepFace.extract_faces(detector_backend="yolov8") calls
(app/src/utils/legacy.py:60-65 on the whole frame, :80-86 in `detect_face`) on the GPU:
YOLOv8n-pose layout, one class, five keypoints (the yolov8n-face.pt file of app/Dockerfile:82-84).

`Detector.detect` returns boxes and keypoints; `Detector.extract_faces` shapes them as
DeepFace's `extract_faces` result entries ({"facial_area": {x, y, w, h, left_eye, right_eye},
"confidence"}) in the order DeepFace's Yolo client produces them, so that it can be registered
with `frhot.quality.install_models(extract_faces=...)`.  Two things DeepFace adds on top are NOT
done here and are named in every result: the anti-spoofing network (`is_real` is reported as
None) and the eye-line alignment of the crop (`face` is the plain box crop).

Weights: no file ships with the reference.  `synthetic_weights` gives random weights of the
architecture for tests and benchmarks; `from_ultralytics_state_dict` folds the batch norms of a
loaded checkpoint's `state_dict()` into this module's layout (to be run where ultralytics and the
file exist)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .weights import pack_entries

REG_MAX, NKPT, IMG = 16, 5, 640


def _convs():
    out = []

    def conv(name, k, cin, cout):
        out.append((name, k, cin, cout))

    def c2f(name, c1, c2, n):
        c = c2 // 2
        conv(f"{name}.cv1", 1, c1, c2)
        for j in range(n):
            conv(f"{name}.m{j}.cv1", 3, c, c)
            conv(f"{name}.m{j}.cv2", 3, c, c)
        conv(f"{name}.cv2", 1, (2 + n) * c, c2)

    conv("m0", 3, 3, 16); conv("m1", 3, 16, 32); c2f("m2", 32, 32, 1); conv("m3", 3, 32, 64); c2f("m4", 64, 64, 2)
    conv("m5", 3, 64, 128); c2f("m6", 128, 128, 2); conv("m7", 3, 128, 256); c2f("m8", 256, 256, 1)
    conv("m9.cv1", 1, 256, 128); conv("m9.cv2", 1, 512, 256)
    c2f("m12", 384, 128, 1); c2f("m15", 192, 64, 1); conv("m16", 3, 64, 64); c2f("m18", 192, 128, 1)
    conv("m19", 3, 128, 128); c2f("m21", 384, 256, 1)
    for l, ch in enumerate((64, 128, 256)):
        conv(f"h.box{l}.0", 3, ch, 64); conv(f"h.box{l}.1", 3, 64, 64); conv(f"h.box{l}.2", 1, 64, 4 * REG_MAX)
        conv(f"h.cls{l}.0", 3, ch, 64); conv(f"h.cls{l}.1", 3, 64, 64); conv(f"h.cls{l}.2", 1, 64, 1)
        conv(f"h.kpt{l}.0", 3, ch, 16); conv(f"h.kpt{l}.1", 3, 16, 16); conv(f"h.kpt{l}.2", 1, 16, 3 * NKPT)
    return out


# (name, kernel size, input channels, output channels) of every convolution, in execution order
CONVS = _convs()
MACS_PER_FRAME = None   # filled below


def _macs():
    hw = {"m0": 320, "m1": 160, "m2": 160, "m3": 80, "m4": 80, "m5": 40, "m6": 40, "m7": 20, "m8": 20, "m9": 20,
          "m12": 40, "m15": 80, "m16": 40, "m18": 40, "m19": 20, "m21": 20}
    total = 0
    for name, k, cin, cout in CONVS:
        if name.startswith("h."):
            side = (80, 40, 20)[int(name[5])]
        else:
            side = hw[name.split(".")[0]]
        total += side * side * k * k * cin * cout
    return total


MACS_PER_FRAME = _macs()       # on the full 640 x 640 canvas


def canvas_of(h: int, w: int):
    """(rows, columns) of the canvas a frame of h x w pixels is detected on: the frame scaled to fit
    640 x 640, padded to the next multiple of 32 in each direction (ultralytics' LetterBox for a frame
    predicted on its own, which is what DeepFace does)."""
    r = min(IMG / h, IMG / w)
    nw, nh = int(round(w * r)), int(round(h * r))
    return nh + (IMG - nh) % 32, nw + (IMG - nw) % 32


def macs_for(h: int, w: int) -> float:
    ch, cw = canvas_of(h, w)
    return MACS_PER_FRAME * (ch * cw) / float(IMG * IMG)


def synthetic_weights(seed: int = 0xFACE, cls_bias: float = -6.0) -> dict:
    """Random weights of the architecture (float32, [cout, k, k, cin]).  LeCun-normal kernels keep
    the activations of the SiLU stack at unit scale; the class branch gets a negative bias so that
    only a few anchors pass the confidence threshold, as with a trained detector."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, k, cin, cout in CONVS:
        std = np.float32(np.sqrt(1.5 / (k * k * cin)))
        out[name + ".w"] = rng.standard_normal(size=(cout, k, k, cin), dtype=np.float32) * std
        out[name + ".b"] = rng.standard_normal(size=(cout,), dtype=np.float32) * np.float32(0.1)
        if name.startswith("h.cls") and name.endswith(".2"):
            out[name + ".b"] += np.float32(cls_bias)
    return out


def pack_blob(weights: dict) -> np.ndarray:
    entries = []
    for name, k, cin, cout in CONVS:
        w = np.ascontiguousarray(weights[name + ".w"]).astype(np.float16)
        b = np.ascontiguousarray(weights[name + ".b"]).astype(np.float32)
        if w.shape != (cout, k, k, cin) or b.shape != (cout,):
            raise ValueError(f"{name}: weight {w.shape} / bias {b.shape}")
        entries.append((name + ".w", 0, w))
        entries.append((name + ".b", 1, b))
    return pack_entries(entries)


def from_ultralytics_state_dict(sd: dict, eps: float = 1e-3) -> dict:
    """`model.model.state_dict()` of the loaded yolov8n-face checkpoint (tensors or arrays; keys
    'model.<i>...conv.weight', '...bn.weight|bias|running_mean|running_var', the heads' last
    convolutions '...weight|bias') -> this module's folded layout.  Module numbers follow the
    YOLOv8 layout: 0-9 backbone, 12/15/18/21 the neck's C2f blocks, 16/19 its strided
    convolutions, 22 the head (cv2 boxes, cv3 classes, cv4 keypoints)."""
    def arr(k):
        v = sd[k]
        return np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)

    def fold(prefix):
        w = arr(prefix + ".conv.weight")                                  # [cout, cin, k, k]
        g, b = arr(prefix + ".bn.weight"), arr(prefix + ".bn.bias")
        mu, var = arr(prefix + ".bn.running_mean"), arr(prefix + ".bn.running_var")
        s = g / np.sqrt(var + eps)
        return (w * s[:, None, None, None]).transpose(0, 2, 3, 1), b - mu * s

    def plain(prefix):
        return arr(prefix + ".weight").transpose(0, 2, 3, 1), arr(prefix + ".bias")

    out = {}
    for name, _k, _cin, _cout in CONVS:
        parts = name.split(".")
        if parts[0] == "h":
            branch, level, idx = parts[1][:3], int(parts[1][3]), int(parts[2])
            mod = {"box": "cv2", "cls": "cv3", "kpt": "cv4"}[branch]
            prefix = f"model.22.{mod}.{level}.{idx}"
            w, b = plain(prefix) if idx == 2 else fold(prefix)
        else:
            i = int(parts[0][1:])
            rest = parts[1:]
            if rest and rest[0].startswith("m") and rest[0] != "m":
                prefix = f"model.{i}.m.{int(rest[0][1:])}.{rest[1]}"
            elif rest:
                prefix = f"model.{i}.{rest[0]}"
            else:
                prefix = f"model.{i}"
            w, b = fold(prefix)
        out[name + ".w"], out[name + ".b"] = np.ascontiguousarray(w), b
    return out


class Detector:
    """A detector on a context's GPU and stream.  detect() waits for its answer."""

    def __init__(self, ctx, blob: np.ndarray, max_batch: int = 8):
        self._lib = _lib.load()
        self._ctx = ctx
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        h = C.c_void_p()
        check(self._lib.frhot_det_create(ctx.handle, blob.ctypes.data_as(C.c_void_p), blob.nbytes, max_batch, C.byref(h)))
        self._h = h
        self.max_batch = max_batch

    def close(self):
        if getattr(self, "_h", None):
            check(self._lib.frhot_det_destroy(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def detect(self, images, conf: float = 0.25, iou: float = 0.7, max_faces: int = 300):
        """images: BGR uint8 arrays [h, w, 3] -> per image a float32 array [n, 20]: x1, y1, x2, y2
        (frame pixels), confidence, five keypoints (x, y, visibility), in descending confidence."""
        from .engine import _crop_args
        out = []
        for i in range(0, len(images), self.max_batch):
            part = images[i:i + self.max_batch]
            keep, ptrs, hw, strides = _crop_args(part)
            n = len(part)
            res = np.zeros((n, max_faces, 20), dtype=np.float32)
            counts = np.zeros(n, dtype=np.int32)
            check(self._lib.frhot_det_detect(self._h, ptrs, hw.ctypes.data_as(C.POINTER(C.c_int32)),
                                             strides.ctypes.data_as(C.POINTER(C.c_int32)), n, conf, iou, max_faces,
                                             res.ctypes.data_as(C.POINTER(C.c_float)),
                                             counts.ctypes.data_as(C.POINTER(C.c_int32))))
            del keep
            out.extend(res[j, :counts[j]].copy() for j in range(n))
        return out

    def debug_read(self, name: str, b: int, h: int, w: int, c: int) -> np.ndarray:
        a = np.empty((b, h, w, c), dtype=np.float16)
        check(self._lib.frhot_det_debug_read(self._h, name.encode("ascii"), b, a.ctypes.data_as(C.c_void_p), a.nbytes))
        return a

    def regions(self, image, conf: float = 0.25):
        """DeepFace's Yolo client, for one frame: [{"x","y","w","h","left_eye","right_eye","confidence"}]."""
        return [region_of(row) for row in self.detect([image], conf=conf)[0]]

    def extract_faces(self, image, expand_percentage: int = 0):
        """The entries DeepFace.extract_faces(detector_backend="yolov8") returns for a frame, without
        the parts named in the module text: `is_real` is None, `face` is the unaligned box crop."""
        return faces_of(image, self.detect([image])[0], expand_percentage)

    def as_extract_faces(self, expand_percentage: int = 10, anti_spoof=None, assume_real: bool = False):
        """The callable frhot.quality.install_models(extract_faces=...) takes, i.e. what the reference's
        detect_face and anti-spoofing pass call (legacy.py:60-65,80-86).  DeepFace judges every face
        with a second network; that network is not part of this package, so the verdict comes from
        `anti_spoof(face crop) -> bool` if one is given, and is True for every face only when the
        operator says so with `assume_real=True`."""
        if anti_spoof is None and not assume_real:
            raise ValueError("no anti-spoofing model: pass anti_spoof=callable, or assume_real=True to accept every face")

        def run(image):
            faces = self.extract_faces(np.ascontiguousarray(image), expand_percentage)
            for f in faces:
                f["is_real"] = bool(anti_spoof(f["face"])) if anti_spoof is not None else True
            return faces
        return run


def faces_of(image, rows, expand_percentage: int = 0) -> list:
    """detect() rows of one frame -> DeepFace.extract_faces entries (facial area enlarged by
    `expand_percentage` as deepface's detection module does, clipped to the frame)."""
    H, W = image.shape[:2]
    out = []
    for reg in (region_of(row) for row in rows):
        x, y, w, h = reg["x"], reg["y"], reg["w"], reg["h"]
        if expand_percentage > 0:
            ew, eh = int(w * expand_percentage / 100), int(h * expand_percentage / 100)
            x, y = max(0, x - ew // 2), max(0, y - eh // 2)
            w, h = min(W - x, w + ew), min(H - y, h + eh)
        out.append({"face": image[y:y + h, x:x + w],
                    "facial_area": {"x": x, "y": y, "w": w, "h": h, "left_eye": reg["left_eye"],
                                    "right_eye": reg["right_eye"]},
                    "confidence": round(reg["confidence"], 2), "is_real": None})
    return out


def region_of(row) -> dict:
    """One detect() row -> DeepFace's facial area: the box through its centre form
    (x = int(cx - w / 2), w = int(w)), keypoint 0 as the right eye and 1 as the left eye, as ints."""
    x1, y1, x2, y2 = (float(v) for v in row[:4])
    w, h = x2 - x1, y2 - y1
    cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
    return {"x": int(cx - w / 2), "y": int(cy - h / 2), "w": int(w), "h": int(h),
            "right_eye": (int(row[5]), int(row[6])), "left_eye": (int(row[8]), int(row[9])),
            "confidence": float(row[4])}
