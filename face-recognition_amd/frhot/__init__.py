"""frhot — MI355X-native recognition hot path behind the reference's own seams.

Host code is Python and only marshals; the arithmetic (preprocess, VGG-Face
descriptor, exact cosine top-k) runs in libfrhot.so, hand-written HIP for gfx950.
"""
from ._lib import EMBED_DIM, FLAG_GAMMA, FLAG_NOFLIP, KEY_ANY, KEY_DEAD, FrhotError  # noqa: F401

__version__ = "0.1.0"
