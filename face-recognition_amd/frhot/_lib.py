"""ctypes binding of libfrhot.so (the C ABI declared in include/frhot.h).

The library is built in-tree (``make -C face-recognition_amd``) and loaded from
this directory.  There is no CPU fallback: if the shared object is missing the
import of anything that needs it raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfrhot.so")

OK, EINVAL, EHIP, ESTATE, ECAPACITY = 0, -1, -2, -3, -4
EMBED_DIM = 4096
FLAG_GAMMA, FLAG_NOFLIP = 1, 2
KEY_ANY, KEY_DEAD = -2, -1

_p = C.c_void_p
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_u8pp = C.POINTER(C.POINTER(C.c_uint8))

# name -> (restype, argtypes); every function of include/frhot.h
SIGNATURES = {
    "frhot_last_error": (C.c_char_p, []),
    "frhot_version": (C.c_char_p, []),
    "frhot_ctx_create": (C.c_int, [C.c_int, _p, C.c_int, C.POINTER(_p)]),
    "frhot_ctx_destroy": (C.c_int, [_p]),
    "frhot_ctx_sync": (C.c_int, [_p]),
    "frhot_model_load": (C.c_int, [_p, _p, C.c_size_t, C.c_float]),
    "frhot_embed": (C.c_int, [_p, _u8pp, _i32p, _i32p, C.c_int, C.c_uint, _f32p]),
    "frhot_embed_dev": (C.c_int, [_p, _p, _i64p, _i32p, _i32p, C.c_int, C.c_uint, _p, _p, C.c_int]),
    "frhot_debug_read": (C.c_int, [_p, C.c_int, C.c_int, _p, C.c_size_t]),
    "frhot_gallery_create": (C.c_int, [_p, C.c_int, C.c_int64, C.POINTER(_p)]),
    "frhot_gallery_destroy": (C.c_int, [_p]),
    "frhot_gallery_upsert": (C.c_int, [_p, _f32p, _i32p, _i64p, C.c_int]),
    "frhot_gallery_tombstone": (C.c_int, [_p, _i64p, C.c_int]),
    "frhot_gallery_fill_synthetic": (C.c_int, [_p, C.c_uint64, C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "frhot_gallery_read": (C.c_int, [_p, _i64p, C.c_int, _p, _i32p]),
    "frhot_gallery_write_f16": (C.c_int, [_p, _p, _i32p, _i64p, C.c_int]),
    "frhot_gallery_rows": (C.c_int64, [_p]),
    "frhot_gallery_search_mode": (C.c_int, [_p, C.c_int]),
    "frhot_search": (C.c_int, [_p, _f32p, _i32p, C.c_int, C.c_int, _f32p, _i64p]),
    "frhot_search_dev": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, _p, _p, C.c_int64]),
    "frhot_search_dev_static": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, _p, _p, C.c_int64]),
    "frhot_search_dev_keyed": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, _p, _p, C.c_int64]),
    "frhot_merge_topk_dev": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, C.c_int, _p, _p]),
    "frhot_merge_topk_packed_dev": (C.c_int, [_p, _p, C.c_int64, C.c_int, C.c_int, C.c_int, _p, _p]),
    "frhot_dist_unique_id": (C.c_int, [_p]),
    "frhot_dist_init": (C.c_int, [_p, C.c_int, C.c_int, _p]),
    "frhot_dist_shutdown": (C.c_int, [_p]),
    "frhot_search_sharded_dev": (C.c_int, [_p, _p, _p, C.c_int, C.c_int, _p, _p, C.c_int64]),
    "frhot_det_create": (C.c_int, [_p, _p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "frhot_det_destroy": (C.c_int, [_p]),
    "frhot_det_detect": (C.c_int, [_p, _u8pp, _i32p, _i32p, C.c_int, C.c_float, C.c_float, C.c_int, _f32p, _i32p]),
    "frhot_det_debug_read": (C.c_int, [_p, C.c_char_p, C.c_int, _p, C.c_size_t]),
    "frhot_recognise": (C.c_int, [_p, _p, _u8pp, _i32p, _i32p, _i32p, C.c_int, C.c_uint, C.c_int,
                                  _f32p, _f32p, _i64p]),
    "frhot_crop_sharpness": (C.c_int, [_p, _u8pp, _i32p, _i32p, C.c_int, _i64p]),
    "frhot_crop_sharpness_dev": (C.c_int, [_p, _p, _i64p, _i32p, _i32p, C.c_int, _p]),
    "frhot_recognise_gated": (C.c_int, [_p, _p, _u8pp, _i32p, _i32p, _i32p, C.c_int, C.c_uint, C.c_int,
                                        _f32p, _f32p, _i64p, _i64p]),
    "frhot_ctx_timing": (C.c_int, [_p, C.c_int]),
    "frhot_ctx_stage_ms": (C.c_int, [_p, _f32p]),
}

_lib = None


class FrhotError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"frhot error {code}: {message}")
        self.code = code
        self.message = message


def _share_torch_hip_runtime() -> None:
    """One process, one HIP runtime.  A PyTorch-ROCm wheel ships its own libamdhip64.so (soname
    libamdhip64.so.7) and asks for it by file name; libfrhot.so asks for the soname.  Loaded
    torch-first, the dynamic loader hands libfrhot.so torch's copy and both share one runtime;
    loaded libfrhot-first, it maps the system copy for us and later torch's copy as well, and
    the second runtime to initialise finds no GPU ("No HIP GPUs are available").  Callers hand
    torch device pointers to this library, so when a torch wheel with a bundled runtime is
    installed and not yet imported, its copy is mapped first (no torch import, nothing else of
    torch is touched); without such a wheel the system runtime serves libfrhot.so alone."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.submodule_search_locations:
        return
    for base in spec.submodule_search_locations:
        path = os.path.join(base, "lib", "libamdhip64.so")
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            return


def load() -> C.CDLL:
    """Load libfrhot.so once and declare every signature.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `make -C face-recognition_amd` "
            "(there is no fallback implementation)")
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise FrhotError(rc, load().frhot_last_error().decode("utf-8", "replace"))
