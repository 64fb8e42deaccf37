"""Oracle: the face detector behind DeepFace's "yolov8" backend, CPU fp32.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the network lives in
ultralytics (YOLOv8n-pose layout, one class, five keypoints: the `yolov8n-face.pt` file DeepFace
0.0.93 downloads, app/Dockerfile:82-84) and the wrapper in deepface
(models/face_detection/Yolo.py); neither is installed here and the reference holds no detection
fixture.  The architecture and the post-processing below are restated from the published
behaviour of those packages; every step that could differ from them is listed in FLAGS.

Reference call sites: app/src/utils/legacy.py:60-65 (anti-spoofing pass on the whole frame) and
:80-86 (`detect_face`), both DeepFace.extract_faces(detector_backend="yolov8", align=True).

Restated network (width 0.25, depth 0.33 of the YOLOv8 family; every "Conv" = convolution without
bias + batch norm + SiLU, folded here into convolution + bias + SiLU):
  0 Conv 3->16 k3 s2 | 1 Conv 16->32 k3 s2 | 2 C2f 32 n1 | 3 Conv 32->64 k3 s2 | 4 C2f 64 n2 |
  5 Conv 64->128 k3 s2 | 6 C2f 128 n2 | 7 Conv 128->256 k3 s2 | 8 C2f 256 n1 | 9 SPPF 256 k5 |
  10 up x2 | 11 cat(10, 6) | 12 C2f 384->128 n1 (no shortcut) | 13 up x2 | 14 cat(13, 4) |
  15 C2f 192->64 n1 | 16 Conv 64->64 k3 s2 | 17 cat(16, 12) | 18 C2f 192->128 n1 |
  19 Conv 128->128 k3 s2 | 20 cat(19, 9) | 21 C2f 384->256 n1 | 22 Pose head on (15, 18, 21):
  box branch Conv c->64 k3, Conv 64->64 k3, conv 64->64 k1 (+bias; 4 sides x 16 bins);
  class branch Conv c->64 k3, Conv 64->64 k3, conv 64->1 k1; keypoint branch Conv c->16 k3,
  Conv 16->16 k3, conv 16->15 k1 (5 points x (x, y, visibility)).
C2f(c1, c2, n): cv1 1x1 c1->c2, split in two halves of c = c2/2, n bottlenecks (3x3 c->c, 3x3 c->c,
plus the input when `shortcut`) each fed by the previous half, cv2 1x1 (2+n)c -> c2 over all halves.
SPPF: cv1 1x1 c->c/2, three chained 5x5/1 max pools, cv2 1x1 over the four maps.

FLAGS (re-verify against ultralytics + deepface where they exist):
  D1 letterbox: ratio min(640/h, 640/w) (enlarging allowed), new size (round(w r), round(h r)),
     bilinear resize with half-pixel centres in float (OpenCV's uint8 fixed-point path differs by
     at most one level), padding 114 split round(d -/+ 0.1).  A frame predicted alone (what DeepFace
     does) is padded only to the next multiple of 32 in each direction ("auto" of ultralytics'
     LetterBox, used when all images of a call share one shape); `auto=False` gives the full
     640 x 640 square ultralytics uses for a call with images of different shapes;
  D2 BGR -> RGB, /255, no mean;
  D3 decode: box distances = expectation over softmax of 16 bins; xywh from anchor centre
     (cell + 0.5) -/+ distances, times the stride; keypoint x = (2 v + cell) * stride with cell
     = anchor - 0.5, visibility through a sigmoid;
  D4 selection: class score > 0.25, greedy IoU 0.7 suppression in score order, at most 300;
  D5 mapping back: subtract the padding, divide by the ratio, clip to the image;
  D6 DeepFace region: x = int(cx - w/2), y = int(cy - h/2), w = int(w), h = int(h); keypoint 0 is the
     RIGHT eye, keypoint 1 the LEFT eye (image order), each kept as ints.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

IMG = 640
STRIDES = (8, 16, 32)
REG_MAX = 16
NKPT = 5


def conv_specs():
    """[(name, k, stride, cin, cout, act)] of every convolution, in execution order."""
    specs = []

    def conv(name, k, s, cin, cout, act=True):
        specs.append((name, k, s, cin, cout, act))

    def c2f(name, c1, c2, n):
        c = c2 // 2
        conv(f"{name}.cv1", 1, 1, c1, c2)
        for j in range(n):
            conv(f"{name}.m{j}.cv1", 3, 1, c, c)
            conv(f"{name}.m{j}.cv2", 3, 1, c, c)
        conv(f"{name}.cv2", 1, 1, (2 + n) * c, c2)

    conv("m0", 3, 2, 3, 16)
    conv("m1", 3, 2, 16, 32)
    c2f("m2", 32, 32, 1)
    conv("m3", 3, 2, 32, 64)
    c2f("m4", 64, 64, 2)
    conv("m5", 3, 2, 64, 128)
    c2f("m6", 128, 128, 2)
    conv("m7", 3, 2, 128, 256)
    c2f("m8", 256, 256, 1)
    conv("m9.cv1", 1, 1, 256, 128)
    conv("m9.cv2", 1, 1, 512, 256)
    c2f("m12", 384, 128, 1)
    c2f("m15", 192, 64, 1)
    conv("m16", 3, 2, 64, 64)
    c2f("m18", 192, 128, 1)
    conv("m19", 3, 2, 128, 128)
    c2f("m21", 384, 256, 1)
    for l, ch in enumerate((64, 128, 256)):
        conv(f"h.box{l}.0", 3, 1, ch, 64)
        conv(f"h.box{l}.1", 3, 1, 64, 64)
        conv(f"h.box{l}.2", 1, 1, 64, 4 * REG_MAX, act=False)
        conv(f"h.cls{l}.0", 3, 1, ch, 64)
        conv(f"h.cls{l}.1", 3, 1, 64, 64)
        conv(f"h.cls{l}.2", 1, 1, 64, 1, act=False)
        conv(f"h.kpt{l}.0", 3, 1, ch, 16)
        conv(f"h.kpt{l}.1", 3, 1, 16, 16)
        conv(f"h.kpt{l}.2", 1, 1, 16, 3 * NKPT, act=False)
    return specs


SHORTCUT = {"m2": True, "m4": True, "m6": True, "m8": True, "m12": False, "m15": False, "m18": False, "m21": False}
C2F_N = {"m2": 1, "m4": 2, "m6": 2, "m8": 1, "m12": 1, "m15": 1, "m18": 1, "m21": 1}


def letterbox(img_bgr: np.ndarray, auto: bool = True):
    """uint8 [h, w, 3] BGR -> (float32 [H, W, 3] RGB in [0, 1], ratio, (pad_left, pad_top)); H, W are
    multiples of 32, at most 640.  FLAGS D1, D2."""
    h, w = img_bgr.shape[:2]
    r = min(IMG / h, IMG / w)
    nw, nh = int(round(w * r)), int(round(h * r))
    dw, dh = IMG - nw, IMG - nh
    if auto:
        dw, dh = dw % 32, dh % 32
    top, left = int(round(dh / 2 - 0.1)), int(round(dw / 2 - 0.1))
    bottom, right = int(round(dh / 2 + 0.1)), int(round(dw / 2 + 0.1))
    src = img_bgr.astype(np.float32)
    if (nh, nw) != (h, w):
        sy, sx = h / nh, w / nw
        fy = (np.arange(nh, dtype=np.float64) + 0.5) * sy - 0.5
        fx = (np.arange(nw, dtype=np.float64) + 0.5) * sx - 0.5
        y0 = np.floor(fy).astype(np.int64)
        x0 = np.floor(fx).astype(np.int64)
        ay = (fy - y0).astype(np.float32)
        ax = (fx - x0).astype(np.float32)
        y0c, y1c = np.clip(y0, 0, h - 1), np.clip(y0 + 1, 0, h - 1)
        x0c, x1c = np.clip(x0, 0, w - 1), np.clip(x0 + 1, 0, w - 1)
        rows0, rows1 = src[y0c], src[y1c]
        t0 = rows0[:, x0c] * (1 - ax)[None, :, None] + rows0[:, x1c] * ax[None, :, None]
        t1 = rows1[:, x0c] * (1 - ax)[None, :, None] + rows1[:, x1c] * ax[None, :, None]
        out = t0 * (1 - ay)[:, None, None] + t1 * ay[:, None, None]
        out = np.floor(out + 0.5)                     # back to whole levels, as a uint8 resize returns
    else:
        out = src
    canvas = np.full((nh + top + bottom, nw + left + right, 3), 114.0, np.float32)
    canvas[top:top + nh, left:left + nw] = out
    return canvas[:, :, ::-1] / np.float32(255.0), r, (left, top)


def _conv(x, w, b, stride, act):
    k = w.shape[1]
    y = F.conv2d(x, torch.from_numpy(np.ascontiguousarray(w)).permute(0, 3, 1, 2), torch.from_numpy(b), stride=stride,
                 padding=k // 2)
    return y * torch.sigmoid(y) if act else y


def forward(x_nhwc: np.ndarray, weights: dict, taps: dict | None = None, calibrate: bool = False):
    """float32 [B, 640, 640, 3] (letterboxed RGB in [0, 1]) -> per level l the raw head maps
    (box [B, h, w, 64], cls [B, h, w, 1], kpt [B, h, w, 15]).  `taps`, when given, receives every
    named intermediate map (NHWC numpy) for per-layer comparisons.  `calibrate` rescales, IN PLACE,
    every convolution's kernel so that its output before the activation has unit spread on this
    input (what batch norm gives a trained network; random kernels otherwise die out or blow up
    within a few blocks), the class branch's last one to 1.5 around its bias."""
    spec = {name: (k, s, act) for name, k, s, _ci, _co, act in conv_specs()}

    def cv(name, x):
        k, s, act = spec[name]
        if calibrate:
            pre = _conv(x, weights[name + ".w"], np.zeros_like(weights[name + ".b"]), s, False)
            target = 1.5 if name.startswith("h.cls") and name.endswith(".2") else 1.0
            weights[name + ".w"] = (weights[name + ".w"] * np.float32(target / max(float(pre.std()), 1e-12))).astype(np.float32)
        return _conv(x, weights[name + ".w"], weights[name + ".b"], s, act)

    def c2f(name, x):
        y = list(cv(f"{name}.cv1", x).chunk(2, 1))
        for j in range(C2F_N[name]):
            t = cv(f"{name}.m{j}.cv2", cv(f"{name}.m{j}.cv1", y[-1]))
            y.append(y[-1] + t if SHORTCUT[name] else t)
        return cv(f"{name}.cv2", torch.cat(y, 1))

    def keep(name, t):
        if taps is not None:
            taps[name] = t.permute(0, 2, 3, 1).numpy()
        return t

    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(x_nhwc, dtype=np.float32)).permute(0, 3, 1, 2)
        x = keep("m0", cv("m0", x))
        x = keep("m1", cv("m1", x))
        x = keep("m2", c2f("m2", x))
        x = keep("m3", cv("m3", x))
        p3 = keep("m4", c2f("m4", x))
        x = keep("m5", cv("m5", p3))
        p4 = keep("m6", c2f("m6", x))
        x = keep("m7", cv("m7", p4))
        x = keep("m8", c2f("m8", x))
        a = cv("m9.cv1", x)
        b1 = F.max_pool2d(a, 5, 1, 2)
        b2 = F.max_pool2d(b1, 5, 1, 2)
        b3 = F.max_pool2d(b2, 5, 1, 2)
        p5 = keep("m9", cv("m9.cv2", torch.cat([a, b1, b2, b3], 1)))
        x = torch.cat([F.interpolate(p5, scale_factor=2, mode="nearest"), p4], 1)
        h4 = keep("m12", c2f("m12", x))
        x = torch.cat([F.interpolate(h4, scale_factor=2, mode="nearest"), p3], 1)
        o3 = keep("m15", c2f("m15", x))
        x = torch.cat([cv("m16", o3), h4], 1)
        o4 = keep("m18", c2f("m18", x))
        x = torch.cat([cv("m19", o4), p5], 1)
        o5 = keep("m21", c2f("m21", x))
        outs = []
        for l, f in enumerate((o3, o4, o5)):
            maps = []
            for br in ("box", "cls", "kpt"):
                t = cv(f"h.{br}{l}.2", cv(f"h.{br}{l}.1", cv(f"h.{br}{l}.0", f)))
                maps.append(t.permute(0, 2, 3, 1).numpy())
            outs.append(tuple(maps))
        return outs


def decode(outs, conf: float = 0.25):
    """Raw head maps -> per image float32 arrays (boxes xywh [n, 4] on the 640 canvas, scores [n],
    keypoints [n, 5, 3] (x, y, visibility)) of the anchors whose score exceeds `conf`, in anchor
    order (level, row, column).  FLAG D3."""
    B = outs[0][0].shape[0]
    per_image = []
    for b in range(B):
        boxes, scores, kpts = [], [], []
        for l, (box, cls, kpt) in enumerate(outs):
            s = STRIDES[l]
            h, w = box.shape[1:3]
            logits = box[b].reshape(h, w, 4, REG_MAX).astype(np.float32)
            e = np.exp(logits - logits.max(-1, keepdims=True))
            dist = (e / e.sum(-1, keepdims=True) * np.arange(REG_MAX, dtype=np.float32)).sum(-1)      # [h, w, 4] l, t, r, b
            ax = np.arange(w, dtype=np.float32)[None, :] + 0.5
            ay = np.arange(h, dtype=np.float32)[:, None] + 0.5
            x1, y1 = ax - dist[..., 0], ay - dist[..., 1]
            x2, y2 = ax + dist[..., 2], ay + dist[..., 3]
            xywh = np.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], -1) * s
            sc = 1.0 / (1.0 + np.exp(-cls[b, ..., 0].astype(np.float32)))
            kp = kpt[b].reshape(h, w, NKPT, 3).astype(np.float32)
            kx = (kp[..., 0] * 2.0 + (ax - 0.5)[..., None]) * s
            ky = (kp[..., 1] * 2.0 + (ay - 0.5)[..., None]) * s
            kv = 1.0 / (1.0 + np.exp(-kp[..., 2]))
            m = sc > conf
            boxes.append(xywh[m])
            scores.append(sc[m])
            kpts.append(np.stack([kx, ky, kv], -1)[m])
        per_image.append((np.concatenate(boxes).astype(np.float32), np.concatenate(scores).astype(np.float32),
                          np.concatenate(kpts).astype(np.float32)))
    return per_image


def iou_xywh(a, b):
    ax1, ay1, ax2, ay2 = a[0] - a[2] / 2, a[1] - a[3] / 2, a[0] + a[2] / 2, a[1] + a[3] / 2
    bx1, by1, bx2, by2 = b[0] - b[2] / 2, b[1] - b[3] / 2, b[0] + b[2] / 2, b[1] + b[3] / 2
    iw, ih = max(0.0, min(ax2, bx2) - max(ax1, bx1)), max(0.0, min(ay2, by2) - max(ay1, by1))
    inter = iw * ih
    return inter / (a[2] * a[3] + b[2] * b[3] - inter + 1e-7)


def nms(boxes, scores, iou: float = 0.7, max_det: int = 300):
    """Greedy suppression in descending score order (ties: lower anchor index first).  FLAG D4."""
    order = sorted(range(len(scores)), key=lambda i: (-float(scores[i]), i))
    kept = []
    for i in order:
        if all(iou_xywh(boxes[i], boxes[j]) <= iou for j in kept):
            kept.append(i)
            if len(kept) == max_det:
                break
    return kept


def regions(img_bgr: np.ndarray, weights: dict, conf: float = 0.25, auto: bool = True):
    """DeepFace-style regions of one image: [{"x","y","w","h","left_eye","right_eye","confidence"}] in
    descending confidence.  FLAGS D5, D6."""
    x, r, (left, top) = letterbox(img_bgr, auto)
    boxes, scores, kpts = decode(forward(x[None], weights), conf)[0]
    H, W = img_bgr.shape[:2]
    out = []
    for i in nms(boxes, scores):
        cx, cy, w, h = (float(v) for v in boxes[i])
        x1 = min(max((cx - w / 2 - left) / r, 0.0), W)
        y1 = min(max((cy - h / 2 - top) / r, 0.0), H)
        x2 = min(max((cx + w / 2 - left) / r, 0.0), W)
        y2 = min(max((cy + h / 2 - top) / r, 0.0), H)
        k = kpts[i].astype(np.float64).copy()
        k[:, 0] = np.clip((k[:, 0] - left) / r, 0, W)
        k[:, 1] = np.clip((k[:, 1] - top) / r, 0, H)
        bw, bh = x2 - x1, y2 - y1
        bx, by = (x1 + x2) / 2, (y1 + y2) / 2
        out.append({"x": int(bx - bw / 2), "y": int(by - bh / 2), "w": int(bw), "h": int(bh),
                    "right_eye": (int(k[0, 0]), int(k[0, 1])), "left_eye": (int(k[1, 0]), int(k[1, 1])),
                    "confidence": float(scores[i])})
    return out
