#!/usr/bin/env python3
"""Headline benchmark: recognised faces/sec = embed + exact top-5 vs a 10M x 512 gallery.

One step = one pass of the hot path over one batch of synthetic crops:
gamma LUT -> resize/pad/normalise -> VGG-Face (13 conv + fc6 + fc7) -> L2 norm ->
exact cosine top-5 against the fp16 gallery.  With N GPUs every rank embeds its
own `--batch` crops (data parallel, no collective), the gallery is row-sharded
(10M/N rows per rank), queries are all-gathered, every shard is scanned with all
N*batch queries, and the per-shard top-k lists are all-gathered and merged.
Per-GPU work is fixed as N grows ("weak" scaling).

Inputs (crops, weights, gallery) are synthetic, generated from fixed seeds, and
resident in HBM before the timed region starts.

Usage: python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "face-recognition_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

GFLOP_PER_FACE_CNN = 2 * 15_466_168_320 / 1e9   # conv stack + fc6 + fc7
HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TFLOPS = 2500.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="crops per GPU per step")
    ap.add_argument("--gallery-rows", type=int, default=10_000_000, help="rows of the whole gallery")
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0xFACE0000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scan64", action="store_true", help="skip the secondary HBM-bound scan measurement")
    ap.add_argument("--cpu-faces", type=int, default=4, help="faces the CPU baseline embeds")
    ap.add_argument("--cpu-rows", type=int, default=400_000, help="gallery rows the CPU baseline scans")
    return ap.parse_args()


def cpu_baseline(ctx, gal, images, args, rows_local):
    """Time the CPU oracle on a bounded sample of the same workload; also gate correctness."""
    from oracle import preprocess as op, search as osr, vggface
    from frhot.weights import synthetic_weights
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    nf = min(args.cpu_faces, len(images))
    w = synthetic_weights(seed=args.seed & 0xFFFF)
    t0 = time.perf_counter()
    x = np.stack([op.preprocess(im, apply_gamma=True) for im in images[:nf]])
    ref = vggface.embed(x, w, threads=threads)
    t_embed = (time.perf_counter() - t0) / nf
    got = ctx.embed(images[:nf], gamma=True)
    cos = (got.astype(np.float64) * ref).sum(1) / np.linalg.norm(got.astype(np.float64), axis=1)
    # scan sample: the first ns rows of this rank's shard, float32 BLAS + partial sort
    ns = int(min(args.cpu_rows, rows_local))
    g16, _ = gal.read(np.arange(ns))
    g32 = g16.astype(np.float32)
    qd = ref[:, :args.dim]
    t0 = time.perf_counter()
    s_ref, r_ref = osr.cosine_topk_fast(qd, g32, args.k)
    t_scan = (time.perf_counter() - t0) / nf * (args.gallery_rows / ns)
    # gate: over the full shard the GPU must return, for every sampled query, scores that are
    # the true dot products of the rows it names and no sampled row may beat its k-th score
    qn = qd / np.linalg.norm(qd, axis=1, keepdims=True)
    s_gpu, r_gpu = gal.search(qn.astype(np.float32), None, args.k)
    rows_back, _ = gal.read(r_gpu.reshape(-1))
    q16 = qn.astype(np.float32).astype(np.float16).astype(np.float32)
    redo = (rows_back.astype(np.float32).reshape(nf, args.k, -1) * q16[:, None, :]).sum(-1)
    ok_scores = bool(np.abs(redo - s_gpu).max() <= 1e-4)
    ok_missed = bool((s_ref[:, 0] <= s_gpu[:, 0] + 1e-4).all() and (s_ref[:, -1] <= s_gpu[:, -1] + 1e-4).all())
    return {
        "value": 1.0 / (t_embed + t_scan), "unit": "faces/s", "cores": threads, "kind": "port",
        "sample": (f"oracle (torch-CPU fp32 VGG-Face + numpy f32 exact top-{args.k}) on {nf} of the bench crops; "
                   f"scan timed on the first {ns} gallery rows and scaled to {args.gallery_rows}"),
        "embed_s_per_face": t_embed, "scan_s_per_face": t_scan,
    }, {"embed_cosine_min": float(cos.min()), "search_scores_ok": ok_scores, "search_no_missed_row": ok_missed}


def pmc_traffic(rows, dim, queries, k):
    """HBM bytes per launch of the scan for this exact workload, from the newest committed PMC
    summary (profiles/*_pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    this bench, FETCH_SIZE doubled for gfx950).  None when no pass covers the workload."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        for ent in doc.get("kernels", {}).values():
            if ent.get("workload") == {"rows": rows, "dim": dim, "queries": queries, "k": k}:
                return ent["traffic_bytes"], f"{doc.get('source', os.path.basename(path))} ({doc.get('correction', '')})"
    return None, None


def main():
    args = parse()
    # Native libraries write to the process's stdout behind Python's back (RCCL prints a version
    # banner when its communicator is created).  Everything this process prints goes to stderr;
    # the one JSON line is written to the original stdout at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1:
        raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nproc-per-node N bench.py ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); none is visible")
    dev_index = local_rank
    if os.environ.get("FRHOT_BENCH_WATCHDOG_S"):      # diagnosis: say where a hung run stands, then end it
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FRHOT_BENCH_WATCHDOG_S"]), exit=True)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # FRHOT_BENCH_REHEARSE_DIST=1 with one rank: a one-rank RCCL group, and the gathers and the merge of the
    # multi-rank step run anyway (what a one-GPU box can rehearse of the N > 1 path)
    rehearse = world == 1 and bool(os.environ.get("FRHOT_BENCH_REHEARSE_DIST"))
    if world > 1 or rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm

    from frhot import synth
    from frhot.dist import ShardedSearch, shard_rows
    from frhot.engine import Context, Gallery
    from frhot.weights import pack_blob, synthetic_weights

    P, B, D, K = world, args.batch, args.dim, args.k
    row_base, rows_local = shard_rows(args.gallery_rows, P, rank)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        ctx = Context(device=dev_index, max_batch=B, stream=stream.cuda_stream)
        ctx.load_model(pack_blob(synthetic_weights(seed=args.seed & 0xFFFF)), input_scale=1.0 / 16.0)
        images = synth.crops(seed=args.seed + 3 + rank, n=B, lo=112, hi=448)
        buf, offsets, hw, strides = synth.pack_crops(images)
        crops_dev = torch.from_numpy(buf).to(dev)
        gal = Gallery(ctx, D, rows_local)
        gal.fill_synthetic(seed=args.seed + 1000 * (rank + 1), row0=0, n=rows_local, nkeys=0, dup_every=100)

        q_local = torch.empty((B, D), dtype=torch.float16, device=dev)
        flags = Context.flags(gamma=True, flip=True)

        def search_local(q_all, s_out, r_out):
            gal.search_dev(q_all.data_ptr(), None, q_all.shape[0], K, s_out.data_ptr(), r_out.data_ptr(), row_base)

        def merge(s_all, r_all, out_s, out_r):
            ctx.merge_topk_dev(s_all.data_ptr(), r_all.data_ptr(), P, P * B, K, out_s.data_ptr(), out_r.data_ptr())

        sharded = ShardedSearch(P, B, D, K, dev, search_local, merge, always_exchange=rehearse)
        last = {}

        def step():
            ctx.embed_dev(crops_dev.data_ptr(), offsets, hw, strides, flags, None, q_local.data_ptr(), D)
            last["s"], last["r"] = sharded(q_local)

        for _ in range(args.warmup):
            step()
        ctx.timing(True)
        if P > 1 or rehearse:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        if P > 1 or rehearse:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        stages = ctx.stage_ms()
        ctx.timing(False)
        if P > 1 or rehearse:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())

        # sanity on the last step's answer: every query found k rows, scores descending
        res_s = last["s"].cpu().numpy()
        res_r = last["r"].cpu().numpy()
        sane = bool((res_r >= 0).all() and (np.diff(res_s, axis=1) <= 0).all() and np.isfinite(res_s).all())

        # Secondary measurement on rank 0: the HBM-bound scan-only shape
        # (1M x 512 fp16 gallery, 64 queries, top-10), timed with the library's stream events.
        scan64 = None
        if rank == 0 and not args.no_scan64:
            n64, bq64, k64 = 1_000_000, 64, 10
            gal64 = Gallery(ctx, D, n64)
            gal64.fill_synthetic(seed=args.seed + 77, row0=0, n=n64, nkeys=0, dup_every=0)
            q64 = q_local[:bq64].contiguous()
            s64 = torch.empty((bq64, k64), dtype=torch.float32, device=dev)
            r64 = torch.empty((bq64, k64), dtype=torch.int64, device=dev)
            for _ in range(3):
                gal64.search_dev(q64.data_ptr(), None, bq64, k64, s64.data_ptr(), r64.data_ptr(), 0)
            ctx.timing(True)
            for _ in range(20):
                gal64.search_dev(q64.data_ptr(), None, bq64, k64, s64.data_ptr(), r64.data_ptr(), 0)
            ms64 = ctx.stage_ms()["scan"]
            ctx.timing(False)
            b64 = n64 * D * 2 + bq64 * D * 2
            tr64, src64 = pmc_traffic(n64, D, bq64, k64)
            scan64 = {"workload": f"scan only: {n64}x{D} fp16 gallery, {bq64} queries, top-{k64}", "bound": "hbm",
                      "achieved": b64 / (ms64 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": b64 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": tr64,
                      "traffic_source": src64,
                      "algorithmic_bytes_per_launch": b64, "ms_per_launch": ms64,
                      "queries_per_s": bq64 / (ms64 * 1e-3)}
            gal64.close()

        line = None
        if rank == 0:
            ms_step = elapsed / args.steps * 1e3
            faces = P * B * args.steps
            # no key filter in this workload: the scan reads the rows and the queries, not the keys
            scan_bytes = rows_local * D * 2 + P * B * D * 2
            traffic, traffic_src = pmc_traffic(rows_local, D, P * B, K)
            scan_gbs = scan_bytes / (stages["scan"] * 1e-3) / 1e9 if stages["scan"] > 0 else 0.0
            scan_tf = 2.0 * P * B * rows_local * D / (stages["scan"] * 1e-3) / 1e12 if stages["scan"] > 0 else 0.0
            cnn_ms = stages["conv"] + stages["fc"]
            cnn_tf = B * GFLOP_PER_FACE_CNN / (cnn_ms * 1e-3) / 1e3 if cnn_ms > 0 else 0.0
            line = {
                "metric": "recognised faces/sec (embed + top-5 vs 10M gallery)",
                "value": faces / elapsed, "unit": "faces/s", "n_gpus": P, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f16", "data": "synthetic",
                "config": {"workload": (f"embed {B} crops/GPU (gamma LUT + VGG-Face + L2) + exact cosine top-{K} vs "
                                        f"{args.gallery_rows}x{D} fp16 gallery row-sharded over {P} GPU(s)"),
                           "batch_per_gpu": B, "gallery_rows": args.gallery_rows, "dim": D, "k": K,
                           "parallelism": f"dp{P}+rowshard{P}"},
                "roofline": {"kernel": "scan_topk_kernel", "bound": "hbm", "achieved": scan_gbs, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": scan_gbs / HBM_PEAK_GBS, "traffic": traffic,
                             "traffic_source": traffic_src,
                             "algorithmic_bytes_per_launch": scan_bytes, "ms_per_launch": stages["scan"],
                             "mfma_tflops": scan_tf, "mfma_frac": scan_tf / MFMA_PEAK_TFLOPS,
                             "note": ("at P*B = %d queries the scan's intensity (~Bq flop/B) sits at the HBM/MFMA "
                                      "ridge: both fractions are reported" % (P * B))},
                "roofline_cnn": {"kernels": "conv1_1 + conv3x3 (patch x3, ring x9) + fc_splitk x2", "bound": "mfma",
                                 "achieved": cnn_tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                 "frac": cnn_tf / MFMA_PEAK_TFLOPS, "gflop_per_face": GFLOP_PER_FACE_CNN,
                                 "ms_per_batch": cnn_ms},
                "stages_ms": stages, "result_sane": sane,
            }
            if scan64 is not None:
                line["roofline_scan_hbm"] = scan64
            if P == 1 and not args.no_cpu_baseline:
                base, gates = cpu_baseline(ctx, gal, images, args, rows_local)
                line["cpu_baseline"] = base
                line["gates"] = gates
        gal.close()
        ctx.close()
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if line is not None:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
