#!/usr/bin/env python3
"""Rehearsal of the sharded serving path on ONE GPU: P ranks (torch.distributed.run, gloo: RCCL
refuses two ranks on one device), each with a frhot context and a gallery shard on GPU 0, serve a
random trace of enrolments (crops embedded on the rank that receives them), evictions past ten
faces per id, deletions and fused recognitions through frhot.sharded_service; rank 0 then replays
the whole trace through ONE GalleryService and compares every answer.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
        --master-port 29531 scripts/sharded_service_rehearsal.py [ticks] > out.json
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "face-recognition_amd")):
    sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from frhot import synth  # noqa: E402
from frhot.engine import Context, Gallery  # noqa: E402
from frhot.gallery_service import GalleryService  # noqa: E402
from frhot.sharded_service import HipShardBackend, ShardedGalleryService  # noqa: E402
from frhot.weights import pack_blob, synthetic_weights  # noqa: E402


def trace(seed, ticks, world, crops):
    """Per tick and rank: (mutations as (kind, crop index or None, fields), recognitions as (collection, crop index, store))."""
    rng = np.random.default_rng(seed)
    cols = ["S1_Employees", "S1_Customers", "S2_Employees"]
    out, clock, used = [], 0, set()
    for t in range(ticks):
        tick = []
        for r in range(world):
            muts, reqs = [], []
            if t == 0 and r == 0:
                muts += [("create_collection", None, c) for c in cols]
            for _ in range(int(rng.integers(0, 4))):
                col = cols[int(rng.integers(0, 3))]
                # every crop is enrolled at most once per collection (under id crop mod 2): two rows with
                # the same descriptor would tie exactly, and the order inside a tie follows row numbers,
                # which differ between one gallery and its shards
                free = [c for c in range(len(crops)) if (col, c) not in used]
                if t > 0 and free and rng.random() < 0.9:
                    ci = free[int(rng.integers(0, len(free)))]
                    used.add((col, ci))
                    clock += 1
                    muts.append(("insert_point", ci,
                                 dict(collection_name=col, id=str(ci % 2), name=f"n{ci % 2}", store_id=col.split("_")[0],
                                      time_created=f"2026/01/01 00:{clock // 60:02d}:{clock % 60:02d}")))
                elif t > 0 and rng.random() < 0.3:
                    muts.append(("delete_point", None, (col, str(int(rng.integers(0, 2))))))
            for _ in range(int(rng.integers(0, 5))):
                reqs.append((cols[int(rng.integers(0, 3))] if rng.random() < 0.9 else "S9_Employees",
                             int(rng.integers(0, len(crops))), "S1" if rng.random() < 0.7 else "S2"))
            tick.append((muts, reqs))
        out.append(tick)
    return out


def main():
    ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cap = 4096
    ctx = Context(device=0, max_batch=16)
    ctx.load_model(pack_blob(synthetic_weights(seed=0xFACE)), input_scale=1.0 / 16.0)
    crops = synth.crops(seed=5, n=36, lo=112, hi=240)
    emb = ctx.embed(crops, gamma=True)              # what an enrolment stores: the crop's descriptor
    svc = ShardedGalleryService(HipShardBackend(ctx, Gallery(ctx, 4096, cap), rank, cap, dev), rank, world, cap, 4096,
                                batch=8, host_staged=True)
    tr = trace(11, ticks, world, crops)

    def ops_of(muts):
        ops = []
        for kind, ci, f in muts:
            if kind == "create_collection":
                ops.append((kind, f))
            elif kind == "insert_point":
                ops.append((kind, dict(f, vector_embedding=emb[ci].tolist())))
            else:
                ops.append((kind, f[0], f[1]))
        return ops

    answers, t0 = [], time.perf_counter()
    for tick in tr:
        muts, reqs = tick[rank]
        a = svc.mutate(ops_of(muts))
        b = svc.recognise_points([(c, crops[ci], s) for c, ci, s in reqs], gamma=True)
        answers.append((a, [(x[0], x[1]) for x in b]))
        if rank == 0 and len(answers) % 5 == 0:
            print(f"tick {len(answers)} of {ticks}", file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    every = [answers]
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, answers)
    shards = sorted({r // cap for r in svc._points})
    counts = {c: svc.point_count(c) for c in svc.get_collections()["collections"]}
    ok = None
    if rank == 0:
        single = GalleryService(ctx, capacity=world * cap)
        mism = nfound = nreq = 0
        for t, tick in enumerate(tr):
            for r, (muts, _q) in enumerate(tick):
                want = []
                for op in ops_of(muts):
                    want.append(single.create_collection(op[1]) if op[0] == "create_collection"
                                else single.insert_point(**op[1]) if op[0] == "insert_point"
                                else single.delete_point(op[1], op[2]))
                if json.dumps(want) != json.dumps(every[r][t][0]):
                    mism += 1
                    print(f"tick {t} rank {r}: mutations answered {every[r][t][0]}, single service {want}", file=sys.stderr)
            for r, (_m, reqs) in enumerate(tick):
                want = single.recognise_points([(c, crops[ci], s) for c, ci, s in reqs], gamma=True)
                for w, g in zip(want, every[r][t][1]):
                    nreq += 1
                    same = w[0] == g[0] and w[1]["message"] == g[1]["message"]
                    if same and w[1].get("data"):
                        nfound += 1
                        # (equal descriptors under several ids tie exactly: compare the score, and the id when it is unique)
                        same = abs(w[1]["data"][0][0] - g[1]["data"][0][0]) < 1e-6
                    if not same:
                        mism += 1
                        print(f"tick {t} rank {r}: recognition answered {g}, single service {w[:2]}", file=sys.stderr)
        ok = mism == 0
        print(json.dumps({"what": "sharded serving rehearsal: ranks share GPU 0, gloo", "world": world, "ticks": ticks,
                          "requests": nreq, "found": nfound, "answers_equal_single_service": ok,
                          "points_per_collection": counts, "shards_holding_points": shards,
                          "seconds": round(dt, 2)}))
        single.close()
    svc.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not ok:
        raise SystemExit("sharded answers differ from the single service's")


if __name__ == "__main__":
    main()
