#!/usr/bin/env python3
"""Developer tool: random enrol / delete / search traffic through GalleryService on the GPU (row
allocator, eviction to ten faces per id, tombstones, search inside one collection by key) against
the CPU restatement of the reference's database facade (oracle.search.Collections).
Usage: python scripts/fuzz_service.py [steps] [seed]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402

from frhot.engine import Context  # noqa: E402
from frhot.gallery_service import GalleryService  # noqa: E402
from oracle import search as osr  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    dim = int(rng.choice([64, 128, 512]))
    ctx = Context(device=0, max_batch=4)
    svc, ref = GalleryService(ctx, capacity=int(rng.choice([8192, 65536])), dim=dim), osr.Collections(dim=dim)
    cols = [f"S{s}_{role}" for s in range(8) for role in ("Employees", "Customers")]
    for c in cols:
        assert svc.create_collection(c) == ref.create_collection(c)
    people = 40
    base = rng.normal(size=(people, dim)).astype(np.float32)
    t = 0
    searches = found = 0
    for step in range(steps):
        op = rng.integers(0, 10)
        col = cols[rng.integers(0, len(cols))]
        store = col.split("_")[0] if rng.random() < 0.9 else "OTHER"
        p = int(rng.integers(0, people))
        pid = f"P{p}"
        vec = base[p] + (0.15 / np.sqrt(dim)) * np.linalg.norm(base[p]) * rng.normal(size=dim).astype(np.float32)
        if op < 5:
            t += 1
            ts = f"2025/01/01 {t // 3600:02d}:{t // 60 % 60:02d}:{t % 60:02d}"
            a = svc.insert_point(col, vec, pid, "n" + pid, store, time_created=ts)
            b = ref.insert_point(col, vec, pid, "n" + pid, store, ts)
            assert a == b, (step, a, b)
        elif op < 6:
            assert svc.delete_point(col, pid) == ref.delete_point(col, pid), step
        else:
            a = svc.search_point(col, vec, store)
            b = ref.search_point(col, vec, store)
            assert a[0] == b[0] and a[1]["message"] == b[1]["message"], (step, a, b)
            searches += 1
            if a[1]["data"]:
                found += 1
                (sa, pa), (sb, pb) = a[1]["data"][0], b[1]["data"][0]
                assert pa["id"] == pb["id"] and abs(sa - sb) < 3e-3, (step, a, b)
        if step % 50 == 0:
            for c in cols:
                assert svc.point_count(c) == len(ref.cols[c]), (step, c)
    print(f"fuzz_service: dim {dim}, {steps} steps, {searches} searches ({found} found) passed")
    svc.close()
    ctx.close()


if __name__ == "__main__":
    main()
