"""Oracle: the check-in quality gates around the recognition path (CPU, numpy).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED for the
third-party steps: no fixture of the reference covers them.

Reference functions this file follows (app/src/utils/legacy.py):
  * check_detect_blur .................. :125-129
  * distance_face_to_camera ............ :113-122
  * eye_aspect_ratio / check_eyes_open . :132-156
  * CalcDistance / DetectDirection ..... :164-188, check_face_left_right :191-208
  * is_full_face ....................... :211-251
  * detect_face (selection loop) ....... :77-103
and the order in which app/src/services/face_service.py:73-219 applies them.
Settings: app/config/settings.py:35-44.

Third-party behaviour restated (opencv-python 4.10.0.84 imgproc):
  * cvtColor(COLOR_BGR2GRAY) on uint8: 15-bit fixed point,
    (3735 B + 19235 G + 9798 R + 16384) >> 15;
  * Laplacian(ddepth=CV_64F), default ksize=1: the 3x3 stencil [[0,1,0],[1,-4,1],[0,1,0]],
    BORDER_REFLECT_101 (index -1 -> 1, n -> n-2; a single row or column reflects onto itself).
The landmark and detection models themselves (MediaPipe FaceMesh / FaceDetection, YOLOv8-face,
FasNet) are not restated: the functions below take their outputs as arrays.
"""
from __future__ import annotations

import math

import numpy as np

KNOWN_FACE_WIDTH = 14.3
EYE_AR_THRESH = 0.2
BLUR_THRESHOLD = 100
LEFT_RIGHT_FACE_THRESHOLD = 2.5
CONF_THRESHOLD = 0.8
LEFT_EYE_LANDMARKS = [33, 160, 158, 133, 153, 144]
RIGHT_EYE_LANDMARKS = [362, 385, 387, 263, 373, 380]


def bgr2gray(img: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) for uint8 [h,w,3|4]."""
    img = np.asarray(img)
    if img.ndim != 3 or img.shape[2] not in (3, 4) or img.dtype != np.uint8:
        raise ValueError(f"BGR2GRAY needs uint8 [h,w,3] or [h,w,4], not {img.dtype} {img.shape}")
    b = img[:, :, 0].astype(np.int64)
    g = img[:, :, 1].astype(np.int64)
    r = img[:, :, 2].astype(np.int64)
    return ((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15).astype(np.uint8)


def _reflect101(n: int) -> np.ndarray:
    """Source index of positions -1 .. n under cv2.BORDER_REFLECT_101 (n == 1: always 0)."""
    idx = np.abs(np.arange(-1, n + 1))
    idx = np.where(idx >= n, 2 * n - 2 - idx, idx)
    return np.clip(idx, 0, n - 1)


def laplacian64(gray: np.ndarray) -> np.ndarray:
    """cv2.Laplacian(gray, cv2.CV_64F) with default aperture and border."""
    gray = np.asarray(gray)
    g = gray[np.ix_(_reflect101(gray.shape[0]), _reflect101(gray.shape[1]))].astype(np.float64)
    return g[:-2, 1:-1] + g[2:, 1:-1] + g[1:-1, :-2] + g[1:-1, 2:] - 4.0 * g[1:-1, 1:-1]


def laplacian_var(img: np.ndarray) -> float:
    """legacy.py:127-128."""
    return float(laplacian64(bgr2gray(img)).var())


def laplacian_sums(img: np.ndarray):
    """Exact (sum, sum of squares) of the Laplacian values (all integers)."""
    lap = laplacian64(bgr2gray(img)).astype(np.int64)
    return int(lap.sum()), int((lap * lap).sum())


def check_detect_blur(img: np.ndarray, threshold=BLUR_THRESHOLD) -> bool:
    """legacy.py:125-129: True = sharp enough."""
    return laplacian_var(img) >= threshold


def distance_face_to_camera(bbox_face, width_or) -> float:
    """legacy.py:113-122."""
    xmin, ymin, xmax, ymax = bbox_face
    P = xmax - xmin
    F_pixel = (4 * width_or) / 4.8
    return (KNOWN_FACE_WIDTH * F_pixel) / P


def _euclid2(a, b) -> float:
    return math.sqrt((a[0] - b[0]) ** 2 + (a[1] - b[1]) ** 2)


def eye_aspect_ratio(eye, lm) -> float:
    """legacy.py:132-141; lm: [468, >=2] normalised landmark coordinates."""
    A = _euclid2(lm[eye[1]], lm[eye[5]])
    B = _euclid2(lm[eye[2]], lm[eye[4]])
    C = _euclid2(lm[eye[0]], lm[eye[3]])
    return (A + B) / (2.0 * C)


def check_eyes_open(faces) -> bool:
    """legacy.py:144-156; faces: list of landmark arrays (None or empty: no face)."""
    if faces:
        lm = faces[0]
        ear = (eye_aspect_ratio(LEFT_EYE_LANDMARKS, lm) + eye_aspect_ratio(RIGHT_EYE_LANDMARKS, lm)) / 2.0
        return ear >= EYE_AR_THRESH
    return False


def detect_direction(lm, threshold=LEFT_RIGHT_FACE_THRESHOLD) -> str:
    """legacy.py:172-188."""
    left = _euclid2(lm[5], lm[234])
    right = _euclid2(lm[5], lm[454])
    result = "straight"
    if left < right:
        if right / left > threshold:
            result = "right"
    elif right < left:
        if left / right > threshold:
            result = "left"
    return result


def check_face_left_right(faces):
    """legacy.py:191-208."""
    if not faces:
        return False, "Face not detected! Please try again"
    direction = detect_direction(faces[0])
    if direction == "left":
        return False, "Face is looking left! Please look straight"
    if direction == "right":
        return False, "Face is looking right! Please look straight"
    return True, "Face is looking straight"


def is_full_face(image_shape, detections):
    """legacy.py:211-251; detections: list of [>=4, 2] relative keypoints
    (right eye, left eye, nose tip, mouth centre, ...)."""
    height, width = image_shape[:2]
    if detections:
        for kp in detections:
            eye_left, eye_right, nose, mouth = kp[1], kp[0], kp[2], kp[3]
            if mouth[0] * width > width or mouth[1] * height > height:
                return False, "Your mouth is not detected! Please show your face"
            if eye_left[0] * width > width or eye_left[1] * height > height:
                return False, "Your left eye is not detected! Please show your face"
            if eye_right[0] * width > width or eye_right[1] * height > height:
                return False, "Your right eye is not detected! Please show your face"
            if nose[0] * width > width or nose[1] * height > height:
                return False, "Your noise is not detected! Please show your face"
        return True, "Face is detected"
    return False, "Face is not detected"


def detect_face(face_detected, image_width):
    """legacy.py:88-103 over the list DeepFace.extract_faces returned: faces are taken in order
    until the first one below the confidence threshold or judged a spoof."""
    boxes, scores, distances = [], [], []
    for f in face_detected:
        if f["confidence"] < CONF_THRESHOLD or not f["is_real"]:
            break
        scores.append(f["confidence"])
        x, y, w, h = list(f["facial_area"].values())[:4]
        distances.append(distance_face_to_camera((x, y, x + w, y + h), image_width))
        boxes.append([x, y, w, h])
    return boxes, np.array(scores), np.array(distances)
