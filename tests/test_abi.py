"""The C-ABI library: it loads, exports every symbol include/frhot.h declares, and fails
loudly (no fallback) where no GPU is present.  No compute calls."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "frhot.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frhot_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = _header_functions()
    for must in ("frhot_ctx_create", "frhot_model_load", "frhot_embed", "frhot_embed_dev", "frhot_gallery_create",
                 "frhot_gallery_upsert", "frhot_gallery_tombstone", "frhot_search", "frhot_search_dev",
                 "frhot_merge_topk_dev", "frhot_recognise", "frhot_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from frhot import _lib
    lib = _lib.load()
    names = _header_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in frhot.h but not exported by libfrhot.so"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signatures and header disagree"
    assert b"gfx950" in lib.frhot_version()


def test_no_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from frhot import FrhotError
    from frhot.engine import Context
    with pytest.raises(FrhotError) as e:
        Context(device=0, max_batch=1)
    assert e.value.code == -2
    from frhot import legacy
    legacy.install_engine(None)
    with pytest.raises(RuntimeError):
        legacy.get_embedding([[[0, 0, 0]]])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "face-recognition_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
