"""The check-in quality gates of the reference's app/src/utils/legacy.py, with the same names,
arguments, return shapes and messages:

  check_detect_blur(img, threshold)            legacy.py:125-129   (GPU: frhot_crop_sharpness)
  distance_face_to_camera(bbox_face, width_or) legacy.py:113-122
  eye_aspect_ratio / check_eyes_open           legacy.py:132-156
  DetectDirection / check_face_left_right      legacy.py:164-208
  is_full_face                                 legacy.py:211-251
  detect_face                                  legacy.py:77-103

The blur measure is pixel arithmetic and runs on the device.  The other gates are a few
operations on the outputs of third-party models (MediaPipe FaceMesh and FaceDetection,
DeepFace's YOLOv8-face detector with its anti-spoofing pass), which this package does not
contain: a host that has them registers callables with `install_models`, and a gate whose
model is not registered raises instead of guessing.
"""
from __future__ import annotations

import numpy as np

KNOWN_FACE_WIDTH = 14.3            # app/config/settings.py:35-40
EYE_AR_THRESH = 0.2
BLUR_THRESHOLD = 100
LEFT_RIGHT_FACE_THRESHOLD = 2.5
CONF_THRESHOLD = 0.8
FACE_EXT = 10                      # per cent the detector's boxes are enlarged by (settings.py:39, legacy.py:84)
LEFT_EYE_LANDMARKS = (33, 160, 158, 133, 153, 144)      # settings.py:43-44
RIGHT_EYE_LANDMARKS = (362, 385, 387, 263, 373, 380)

_engine = None          # an object with .crop_sharpness(list of crops) -> float64 [n]
_face_mesh = None       # BGR image -> list of faces, each [468, >=2] normalised landmark coordinates
_face_detection = None  # BGR image -> list of detections, each [6, 2] relative keypoints
_extract_faces = None   # BGR image -> list of {"confidence", "is_real", "facial_area": {x, y, w, h, ...}}


def install_engine(engine) -> None:
    """Register the device context check_detect_blur measures with (a frhot.engine.Context)."""
    global _engine
    _engine = engine


def install_models(face_mesh=None, face_detection=None, extract_faces=None) -> None:
    """Register the third-party models behind the landmark and detection gates.
    face_mesh: what mp.solutions.face_mesh.FaceMesh(0.5, 0.5).process gives, as arrays;
    face_detection: mp.solutions.face_detection.FaceDetection(0, 0.5) relative keypoints;
    extract_faces: DeepFace.extract_faces(detector_backend="yolov8", align=True,
    expand_percentage=10, anti_spoofing=True)."""
    global _face_mesh, _face_detection, _extract_faces
    _face_mesh, _face_detection, _extract_faces = face_mesh, face_detection, extract_faces


def installed_models() -> dict:
    """The registered callables themselves (None where nothing is registered)."""
    return {"face_mesh": _face_mesh, "face_detection": _face_detection, "extract_faces": _extract_faces}


def models_installed() -> dict:
    return {"face_mesh": _face_mesh is not None, "face_detection": _face_detection is not None,
            "extract_faces": _extract_faces is not None, "engine": _engine is not None}


def _need(model, what):
    if model is None:
        raise RuntimeError(f"frhot.quality: no {what} is installed (install_models / install_engine)")
    return model


def laplacian_variance_of(img) -> float:
    """cv2.Laplacian(cv2.cvtColor(img, COLOR_BGR2GRAY), CV_64F).var(), measured on the GPU."""
    img = np.asarray(img)
    if img.ndim != 3 or img.shape[2] not in (3, 4):
        raise ValueError(f"BGR2GRAY needs a 3- or 4-channel image, not {img.shape}")   # cv2.cvtColor raises too
    crop = np.ascontiguousarray(img[:, :, :3], dtype=np.uint8)
    return float(_need(_engine, "device context").crop_sharpness([crop])[0])


def check_detect_blur(img, threshold=BLUR_THRESHOLD) -> bool:
    """True when the image is sharp enough (legacy.py:125-129)."""
    return laplacian_variance_of(img) >= threshold


def distance_face_to_camera(bbox_face, width_or) -> float:
    """Pinhole estimate in cm from the face box's width: a 4 mm lens over a 4.8 mm sensor and a
    14.3 cm face (legacy.py:113-122)."""
    xmin, _ymin, xmax, _ymax = bbox_face
    focal_px = (4 * width_or) / 4.8
    return (KNOWN_FACE_WIDTH * focal_px) / (xmax - xmin)


def _xy(landmarks, idx) -> np.ndarray:
    return np.asarray([[landmarks[i][0], landmarks[i][1]] for i in idx], dtype=np.float64)


def eye_aspect_ratio(eye_landmarks, face_landmarks) -> float:
    """(|p1 p5| + |p2 p4|) / (2 |p0 p3|) over the six eye contour points (legacy.py:132-141)."""
    p = _xy(face_landmarks, eye_landmarks)
    opening = np.hypot(*(p[1] - p[5])) + np.hypot(*(p[2] - p[4]))
    return float(opening / (2.0 * np.hypot(*(p[0] - p[3]))))


def check_eyes_open(img_decode) -> bool:
    """Mean aspect ratio of both eyes of the first face >= 0.2; no face: False (legacy.py:144-156)."""
    faces = _need(_face_mesh, "face-mesh model")(img_decode)
    if not faces:
        return False
    lm = faces[0]
    ear = (eye_aspect_ratio(LEFT_EYE_LANDMARKS, lm) + eye_aspect_ratio(RIGHT_EYE_LANDMARKS, lm)) / 2.0
    return ear >= EYE_AR_THRESH


def DetectDirection(landmark, threshold=LEFT_RIGHT_FACE_THRESHOLD) -> str:
    """Nose tip (5) against the two face edges (234, 454): the larger over the smaller distance
    above `threshold` means a turned head (legacy.py:172-188)."""
    nose, edge_l, edge_r = _xy(landmark, (5, 234, 454))
    left, right = np.hypot(*(nose - edge_l)), np.hypot(*(nose - edge_r))
    if left < right and right / left > threshold:
        return "right"
    if right < left and left / right > threshold:
        return "left"
    return "straight"


def check_face_left_right(img_decode):
    """(ok, message) of the head-direction gate (legacy.py:191-208)."""
    faces = _need(_face_mesh, "face-mesh model")(img_decode)
    if not faces:
        return False, "Face not detected! Please try again"
    direction = DetectDirection(faces[0])
    if direction != "straight":
        return False, f"Face is looking {direction}! Please look straight"
    return True, "Face is looking straight"


# the four keypoints is_full_face looks at, in its order, with the word its message uses
_FULL_FACE_PARTS = ((3, "mouth"), (1, "left eye"), (0, "right eye"), (2, "noise"))


def is_full_face(image):
    """(ok, message): every detection's mouth, eyes and nose tip lie inside the crop
    (legacy.py:211-251; the messages, "noise" included, are the reference's)."""
    height, width = image.shape[:2]
    detections = _need(_face_detection, "face-detection model")(image)
    if not detections:
        return False, "Face is not detected"
    for kp in detections:
        for idx, word in _FULL_FACE_PARTS:
            if kp[idx][0] * width > width or kp[idx][1] * height > height:
                return False, f"Your {word} is not detected! Please show your face"
    return True, "Face is detected"


def detect_face(image):
    """boxes [[x, y, w, h]], scores, distances of the faces the detector returned, in its order,
    up to the first one below the confidence threshold or judged a spoof (legacy.py:77-103)."""
    boxes, scores, distances = [], [], []
    for found in _need(_extract_faces, "face detector")(image):
        if found["confidence"] < CONF_THRESHOLD or not found["is_real"]:
            break
        x, y, w, h = list(found["facial_area"].values())[:4]
        boxes.append([x, y, w, h])
        scores.append(found["confidence"])
        distances.append(distance_face_to_camera((x, y, x + w, y + h), image.shape[1]))
    return boxes, np.array(scores), np.array(distances)
