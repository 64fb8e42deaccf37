"""CPU oracle for the recognition hot path (TEST INFRASTRUCTURE ONLY).

This package restates, on the CPU, the arithmetic of the reference's recognition
path: crop -> gamma LUT -> DeepFace "VGG-Face" preprocessing -> VGG-16 forward to
post-ReLU fc7 -> L2 normalisation -> exact cosine top-k with a collection/store
filter -> threshold + id vote.

PARITY UNPINNED: the reference ships no golden vectors, known-answer tests or
numeric fixtures for this path, and the packages that own the arithmetic
(deepface 0.0.93, tensorflow 2.13, opencv-python 4.10.0.84, qdrant 1.14.1) are
not installed here.  The third-party steps are restated from the published
behaviour of those pinned versions; the steps that live in the reference's own
files are restated from those files and cite them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker / the timed CPU baseline.  The product
(the frhot package and libfrhot.so) never imports it.
"""
