"""Writes the small fixtures under tests/golden/.

None of them comes from running the reference (its packages are not installed
here and it ships no numeric fixtures): the gamma table evaluates the formula of
app/src/utils/legacy.py:108-109 directly, the vote cases are worked by hand from
qdrant_database_FE/app.py:340-384, two blur-gate cases are worked by hand from the
Laplacian stencil, and the preprocessing / search / blur vectors are
outputs of this repository's own CPU oracle, kept so that a change of the oracle
shows up as a diff ("parity unpinned" for those).

Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "face-recognition_amd"))


def main():
    inv = 1.0 / 1.5
    lut = np.array([(i / 255.0) ** inv * 255 for i in np.arange(0, 256)]).astype("uint8")
    np.save(os.path.join(HERE, "gamma_lut_1p5.npy"), lut)

    # hits are (score, id) in descending score; want = (score, id) or None; threshold 0.54
    cases = [
        {"hits": [[.9, "A"], [.8, "A"], [.7, "B"], [.5, "B"], [.4, "C"]], "want": [.9, "A"]},   # majority
        {"hits": [[.9, "A"], [.8, "B"], [.3, "C"]], "want": [.9, "A"]},                         # all counts equal
        {"hits": [[.9, "A"], [.8, "B"], [.7, "B"], [.6, "C"], [.55, "C"]], "want": [.8, "B"]},  # first max count
        {"hits": [[.6, "A"], [.58, "B"], [.57, "B"], [.2, "B"]], "want": [.58, "B"]},           # max over all B hits
        {"hits": [[.56, "B"], [.55, "A"], [.53, "B"]], "want": [.56, "B"]},                     # equal after filter
        {"hits": [[.6, "A"]], "want": [.6, "A"]},                                               # single id
        {"hits": [[.53, "A"], [.5, "B"]], "want": None},                                        # none kept
        {"hits": [[.9, "A"], [.2, "A"]], "want": [.9, "A"]},
        {"hits": [], "want": None},
        {"hits": [[.54, "A"]], "want": [.54, "A"]},                                             # >= is inclusive
        {"hits": [[.539999, "A"]], "want": None},
        {"hits": [[.9, "A"], [.9, "B"]], "want": [.9, "A"]},                                    # tie: first maximum
        {"hits": [[.7, "A"], [.65, "B"], [.6, "A"], [.58, "B"], [.56, "C"]], "want": [.7, "A"]},  # A2 B2 C1: first max
        {"hits": [[.8, "A"], [.7, "B"], [.6, "C"], [.55, "D"], [.545, "E"]], "want": [.8, "A"]},
    ]
    with open(os.path.join(HERE, "vote_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)

    from frhot import synth
    from oracle import preprocess as op, search as osr
    rng = np.random.default_rng(7)
    crops = [synth.crop(rng, 37, 53), synth.crop(rng, 90, 60), synth.crop(rng, 28, 28)]
    out = {}
    for i, c in enumerate(crops):
        out[f"crop{i}"] = c
        out[f"u8_{i}"] = op.preprocess_u8(c, apply_gamma=True, flip=True)
    np.savez_compressed(os.path.join(HERE, "preprocess_oracle.npz"), **out)

    g = synth.unit_gallery(seed=5, n=300, d=64)
    g[200] = g[100]
    q = synth.planted_queries(np.random.default_rng(6), g[[3, 100, 250]], 0.8)
    keys = (np.arange(300) % 3).astype(np.int32)
    s, r = osr.cosine_topk(q, g, 5, keys=keys, qkeys=np.array([0, -2, 1], np.int32))
    np.savez_compressed(os.path.join(HERE, "search_oracle.npz"), g=g.astype(np.float16), q=q, keys=keys,
                        qkeys=np.array([0, -2, 1], np.int32), scores=s, rows=r)
    # blur gate: two cases worked by hand from the stencil and the border rule, then oracle outputs
    from oracle import quality as oq
    hand = [
        # grey 3x3 = [[0,0,0],[0,10,0],[0,0,0]] (B=G=R).  Row -1 and row 3 mirror row 1 (and so the
        # columns), so an edge pixel next to the centre sees the 10 twice:
        # Laplacian = [[0,20,0],[20,-40,20],[0,20,0]]
        {"grey": [[0, 0, 0], [0, 10, 0], [0, 0, 0]], "sum": 40, "sumsq": 3200},
        # one row [0, 10, 30]: up = down = centre (a single row reflects onto itself); left/right reflect
        # over the end pixels: lap = [2*0 + 10 + 10 - 0, 2*10 + 0 + 30 - 40, 2*30 + 10 + 10 - 120] = [20, 10, -40]
        {"grey": [[0, 10, 30]], "sum": -10, "sumsq": 2100},
    ]
    with open(os.path.join(HERE, "blur_hand_cases.json"), "w") as f:
        json.dump(hand, f, indent=1)
    rng = np.random.default_rng(9)
    out = {}
    for i, (h, w) in enumerate([(37, 53), (64, 64), (33, 129), (1, 70), (90, 1)]):
        c = synth.crop(rng, h, w)
        out[f"crop{i}"] = c
        out[f"sums{i}"] = np.array(oq.laplacian_sums(c), np.int64)
        out[f"var{i}"] = np.float64(oq.laplacian_var(c))
    np.savez_compressed(os.path.join(HERE, "blur_oracle.npz"), **out)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
