"""Drop-in for the hot-path functions of the reference's app/src/utils/legacy.py.

Same names, arguments and return shapes:
  get_embedding(imgf, imgf_real) -> (list of 4096 floats, unit norm; is_real)
  adjust_gamma(image, gamma)     -> uint8 image through the power-law LUT
  check_condition(data, is_checkin) -> (ok, message)

get_embedding runs on the GPU through libfrhot.so (via the process-wide engine
installed with `install_engine`); there is no CPU path.  The reference derives
`is_real` from a YOLOv8 + anti-spoofing pass over the whole image (legacy.py:60-74);
those models are not part of this package: the verdict comes from the detector a host
registered with frhot.quality.install_models, and is True when there is none.
"""
from __future__ import annotations

import numpy as np

_engine = None  # an object with .embed(crop, gamma=False) -> np.ndarray [4096]


def install_engine(engine) -> None:
    """Register the embedder get_embedding uses (a frhot.batcher.EmbedBatcher or compatible)."""
    global _engine
    _engine = engine


def get_embedding(imgf, imgf_real=None):
    """legacy.py:48-74.  imgf: uint8 HxWx3 BGR crop (already gamma-adjusted by the caller)."""
    if _engine is None:
        raise RuntimeError("frhot.legacy.install_engine() has not been called: no GPU embedder is available")
    imgf = np.asarray(imgf)
    if imgf.ndim != 3:
        # DeepFace.represent(detector_backend="skip") raises for non-3-dimensional input
        raise ValueError(f"Input img must be 3 dimensional but it is {imgf.shape}")
    if imgf.shape[2] != 3:
        # a 4-channel crop (cv2.imdecode(flags=-1) of an RGBA file) reaches the Keras model as
        # [1,224,224,4] in the reference and fails there: same outcome, the caller answers 500
        raise ValueError(f"Input img must have 3 channels but it is {imgf.shape}")
    emb = _engine.embed(np.ascontiguousarray(imgf, dtype=np.uint8), gamma=False)
    return [float(v) for v in emb], face_is_real(imgf_real)


def face_is_real(imgf_real) -> bool:
    """legacy.py:60-74: the anti-spoofing verdict of the most confident face of the whole image
    (of the first one when only one was found)."""
    from . import quality
    if imgf_real is None or not quality.models_installed()["extract_faces"]:
        return True
    faces = quality._extract_faces(imgf_real)
    if len(faces) == 0:
        # DeepFace.extract_faces raises when it finds no face (enforce_detection defaults to True) and
        # the reference indexes the list unguarded; either way the caller answers 500
        raise ValueError("Face could not be detected in the whole image")
    best = max(range(len(faces)), key=lambda i: faces[i]['confidence']) if len(faces) > 1 else 0
    return bool(faces[best]["is_real"])


def gamma_table(gamma: float = 1.0) -> np.ndarray:
    """legacy.py:108-109: float64 power, truncating cast."""
    inv = 1.0 / gamma
    return np.array([(i / 255.0) ** inv * 255 for i in np.arange(0, 256)]).astype("uint8")


def adjust_gamma(image, gamma=1.0):
    """legacy.py:106-110 (cv2.LUT == per-byte table lookup)."""
    return gamma_table(gamma)[np.asarray(image, dtype=np.uint8)]


def check_condition(data, is_checkin=True):
    """legacy.py:278-292."""
    if is_checkin is False:
        if data.id is None or data.name is None or data.id == "" or data.name == "":
            return False, "id and name are required"
    if len(data.img_base64) == 0:
        return False, "invalid"
    if data.role != '1' and data.role != '0':
        return False, "invalid"
    if data.store_id is None or data.store_id == "":
        return False, "store_id is required"
    return True, "Success"
