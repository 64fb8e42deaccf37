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


# Scan-only shapes beside the headline step: name -> (gallery rows, dim, queries, k, label, bound, launches).
# rank8 is what ONE rank scans per step when the driver runs this bench on 8 GPUs (10 M / 8 rows against
# all 8 x 256 gathered queries: MFMA-bound); 10m_q64 is the headline gallery at a serving-sized batch.
SCAN_SHAPES = {
    "config2": (1_000_000, 512, 64, 10, "BASELINE config 2 (HBM-bound)", "hbm", 20),
    "rank8": (1_250_000, 512, 2048, 5, "one rank's scan at 8 GPUs: 10 M / 8 rows x 8 * 256 queries (MFMA-bound)", "mfma", 10),
    "10m_q64": (10_000_000, 512, 64, 5, "the headline gallery at a serving-sized batch of queries (HBM-bound)", "hbm", 10),
}


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
    ap.add_argument("--no-scan64", action="store_true", help="skip the secondary scan-only measurements")
    ap.add_argument("--no-scan4096", action="store_true", help="skip the 1M x 4096 scan-only shape (profiling passes: "
                    "it shares its kernel with the headline scan)")
    ap.add_argument("--no-scan-extra", action="store_true", help="skip the rank8 and 10m_q64 scan-only shapes "
                    "(profiling passes: they share kernels with other workloads)")
    ap.add_argument("--no-detector", action="store_true", help="skip the face detector's throughput measurement")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the PCIe-inclusive (host crops in) measurement")
    ap.add_argument("--scan-shape-only", choices=sorted(SCAN_SHAPES), default=None,
                    help="run only this scan-only shape, a few launches, and print its object (counter passes: "
                         "a kernel name then stands for one workload)")
    ap.add_argument("--cpu-faces", type=int, default=4, help="faces the CPU baseline embeds")
    ap.add_argument("--cpu-rows", type=int, default=400_000, help="gallery rows the CPU baseline scans")
    return ap.parse_args()


def host_cpu_info():
    """CPU model, physical cores and logical CPUs of the host, from /proc/cpuinfo."""
    model, cores, logical = None, set(), 0
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            key, _, val = ln.partition(":")
            key, val = key.strip(), val.strip()
            if key == "processor":
                logical += 1
            elif key == "model name" and model is None:
                model = val
            elif key == "physical id":
                phys = val
            elif key == "core id":
                core = val
            elif not key and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or None), (logical or os.cpu_count())


def cpu_threads():
    """Threads the CPU baseline may use: the share this job was given (OMP_NUM_THREADS on the GPU
    box), else every CPU the process may run on."""
    n = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if n > 0:
        return n
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def cpu_baseline(ctx, gal, images, args, rows_local, blur_sums=None):
    """Time the CPU oracle on a bounded sample of the same workload (all threads of this job's CPU
    share, then one thread); also gate correctness of the GPU answer on the same sample.
    blur_sums: the blur-gate leg's device answer for `images`, checked on a few of them."""
    from oracle import preprocess as op, quality as oq, search as osr, vggface
    from frhot.weights import synthetic_weights
    threads = cpu_threads()
    nf = min(args.cpu_faces, len(images))
    w = synthetic_weights(seed=args.seed & 0xFFFF)
    ns = int(min(args.cpu_rows, rows_local))
    g16, _ = gal.read(np.arange(ns))
    g32 = g16.astype(np.float32)

    def run(nthreads, faces):
        torch.set_num_threads(nthreads)
        try:
            from threadpoolctl import threadpool_limits
            limit = threadpool_limits(limits=nthreads)
        except ImportError:
            limit = None
        try:
            t0 = time.perf_counter()
            x = np.stack([op.preprocess(im, apply_gamma=True) for im in images[:faces]])
            ref = vggface.embed(x, w, threads=nthreads)
            t_embed = (time.perf_counter() - t0) / faces
            qd = ref[:, :args.dim]
            t0 = time.perf_counter()
            s_ref, r_ref = osr.cosine_topk_fast(qd, g32, args.k)
            t_scan = (time.perf_counter() - t0) / faces * (args.gallery_rows / ns)
        finally:
            if limit is not None:
                limit.restore_original_limits()
        return ref, s_ref, t_embed, t_scan

    ref, s_ref, t_embed, t_scan = run(threads, nf)
    _, _, t_embed1, t_scan1 = run(1, 1)
    torch.set_num_threads(threads)
    got = ctx.embed(images[:nf], gamma=True)
    cos = (got.astype(np.float64) * ref).sum(1) / np.linalg.norm(got.astype(np.float64), axis=1)
    # gate: over the full shard the GPU must return, for every sampled query, scores that are
    # the true dot products of the rows it names and no sampled row may beat its k-th score
    qd = ref[:, :args.dim]
    qn = qd / np.linalg.norm(qd, axis=1, keepdims=True)
    s_gpu, r_gpu = gal.search(qn.astype(np.float32), None, args.k)
    rows_back, _ = gal.read(r_gpu.reshape(-1))
    q16 = qn.astype(np.float32).astype(np.float16).astype(np.float32)
    redo = (rows_back.astype(np.float32).reshape(nf, args.k, -1) * q16[:, None, :]).sum(-1)
    ok_scores = bool(np.abs(redo - s_gpu).max() <= 1e-4)
    ok_missed = bool((s_ref[:, 0] <= s_gpu[:, 0] + 1e-4).all() and (s_ref[:, -1] <= s_gpu[:, -1] + 1e-4).all())
    gates = {"embed_cosine_min": float(cos.min()), "search_scores_ok": ok_scores, "search_no_missed_row": ok_missed}
    if blur_sums is not None:
        sample = sorted({0, 1, len(images) // 2, len(images) - 1})
        gates["blur_sums_equal_oracle"] = all(
            tuple(int(v) for v in blur_sums[i]) == oq.laplacian_sums(images[i]) for i in sample)
    model, phys, logical = host_cpu_info()
    return {
        "value": 1.0 / (t_embed + t_scan), "unit": "faces/s", "cores": threads, "kind": "port",
        "sample": (f"oracle (torch-CPU fp32 VGG-Face + numpy f32 exact top-{args.k}) on {nf} of the bench crops; "
                   f"scan timed on the first {ns} gallery rows and scaled to {args.gallery_rows}; "
                   f"value_1thread: one crop and the same row slice on one thread"),
        "embed_s_per_face": t_embed, "scan_s_per_face": t_scan,
        "value_1thread": 1.0 / (t_embed1 + t_scan1), "embed_s_per_face_1thread": t_embed1,
        "scan_s_per_face_1thread": t_scan1,
        "threads": threads, "cpu_model": model, "physical_cores": phys, "logical_cpus": logical,
    }, gates


def keyed_search_leg(gal, n, d, run, bq=64, k=5, iters=20):
    """Host-call searches (frhot_search: queries in, answers out) of `bq` queries that each look inside
    a collection of their own (`run` rows stored together), over the whole gallery and by key."""
    rng = np.random.default_rng(17)
    nkeys = n // run
    qkeys = rng.choice(nkeys, size=bq, replace=False).astype(np.int32)
    rows = qkeys.astype(np.int64) * run + rng.integers(0, run, size=bq)
    q16, rk = gal.read(rows)
    q = q16.astype(np.float32)
    res = {}
    for name, mode in (("whole_gallery", gal.SEARCH_WHOLE), ("by_key", gal.SEARCH_BY_KEY)):
        gal.search_mode(mode)
        for _ in range(3):
            s, r = gal.search(q, qkeys, k)
        t0 = time.perf_counter()
        for _ in range(iters):
            s, r = gal.search(q, qkeys, k)
        res[name] = ((time.perf_counter() - t0) / iters * 1e3, s, r)
    gal.search_mode(gal.SEARCH_AUTO)
    return {"workload": f"frhot_search, {bq} queries in {bq} collections of {run} rows, {n}x{d} gallery of {nkeys} "
                        f"collections, top-{k}, host queries in and host answers out",
            "ms_whole_gallery": res["whole_gallery"][0], "ms_by_key": res["by_key"][0],
            "same_answers": bool(np.array_equal(res["whole_gallery"][2], res["by_key"][2])
                                 and np.array_equal(res["whole_gallery"][1], res["by_key"][1])),
            "top1_is_the_asked_row": bool((res["by_key"][2][:, 0] == rows).all() and (rk == qkeys).all())}


def scan_only(ctx, Gallery, dev, n, d, seed, shapes, keyed_run=0, gal=None):
    """Scan-only shapes on one gallery of their own (or on `gal`, n x d, which stays open), timed with
    the library's stream events.
    shapes: [(queries, k, label, bound, launches)]; returns one roofline object per shape (and, with
    keyed_run, the gallery's rows carry keys in runs of that many and a keyed-search object follows)."""
    own = gal is None
    if own:
        gal = Gallery(ctx, d, n)
        gal.fill_synthetic(seed=seed, row0=0, n=n, nkeys=-keyed_run, dup_every=0)
    outs = []
    for bq, k, label, bound, iters in shapes:
        rows16, _ = gal.read(np.arange(bq) * (n // bq))
        q = torch.from_numpy(rows16).to(dev)                    # stored rows as queries: unit fp16
        s_out = torch.empty((bq, k), dtype=torch.float32, device=dev)
        r_out = torch.empty((bq, k), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for _ in range(3):
            gal.search_dev(q.data_ptr(), None, bq, k, s_out.data_ptr(), r_out.data_ptr(), 0)
        ctx.timing(True)
        for _ in range(iters):
            gal.search_dev(q.data_ptr(), None, bq, k, s_out.data_ptr(), r_out.data_ptr(), 0)
        ms = ctx.stage_ms()["scan"]
        ctx.timing(False)
        # (a gallery with planted duplicates: row 100 i + 99 repeats row 100 i + 98, the lower row comes first)
        asked = np.arange(bq) * (n // bq)
        ok = bool(np.isin(r_out[:, 0].cpu().numpy() - asked, (0, -1)).all()
                  and (r_out[:, 0].cpu().numpy() == asked).mean() >= 0.9)
        nbytes = n * d * 2 + bq * d * 2
        gbs = nbytes / (ms * 1e-3) / 1e9
        tf = 2.0 * bq * n * d / (ms * 1e-3) / 1e12
        tr, src = pmc_traffic(n, d, bq, k)
        outs.append({"workload": f"scan only: {n}x{d} fp16 gallery, {bq} queries, top-{k}", "bound": bound,
                     "achieved": gbs if bound == "hbm" else tf,
                     "peak": HBM_PEAK_GBS if bound == "hbm" else MFMA_PEAK_TFLOPS,
                     "unit": "GB/s" if bound == "hbm" else "TFLOP/s",
                     "frac": gbs / HBM_PEAK_GBS if bound == "hbm" else tf / MFMA_PEAK_TFLOPS,
                     "hbm_gbs": gbs, "hbm_frac": gbs / HBM_PEAK_GBS, "mfma_tflops": tf,
                     "mfma_frac": tf / MFMA_PEAK_TFLOPS, "traffic": tr, "traffic_source": src,
                     "algorithmic_bytes_per_launch": nbytes, "ms_per_launch": ms,
                     "queries_per_s": bq / (ms * 1e-3), "top1_is_the_asked_row": ok, "label": label})
    if keyed_run:
        outs.append(keyed_search_leg(gal, n, d, keyed_run))
    if own:
        gal.close()
    return outs


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


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')} rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def detector_leg(ctx, args):
    """The face detector in front of the recognition path (not part of `value`): 32 frames of 480 x 640 as
    host arrays through frhot_det_detect (letterbox, YOLOv8n-pose forward, decode, suppression), random
    weights of the architecture, host wall clock."""
    from frhot import detector as D
    from frhot import synth
    nb = 32
    det = D.Detector(ctx, D.pack_blob(D.synthetic_weights(3)), max_batch=nb)
    try:
        frames = [np.ascontiguousarray(f[:480, :640]) for f in synth.crops(seed=args.seed + 31, n=nb, lo=640, hi=700)]
        for _ in range(2):
            det.detect(frames)
        n_it = 10
        t0 = time.perf_counter()
        for _ in range(n_it):
            det.detect(frames)
        dt = (time.perf_counter() - t0) / n_it
    finally:
        det.close()
    tf = 2.0 * D.macs_for(480, 640) * nb / dt / 1e12
    return {"what": "frhot_det_detect: 32 host frames of 480x640 -> face regions (YOLOv8n-pose layout, random weights)",
            "frames_per_s": nb / dt, "ms_per_call": dt * 1e3, "frames": nb, "canvas": list(D.canvas_of(480, 640)),
            "gflop_per_frame": 2.0 * D.macs_for(480, 640) / 1e9,
            "bound": "mfma", "achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS}


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
    # Rehearsal switches (never set by the driver): FRHOT_BENCH_SHARE_GPU=1 puts every rank on GPU 0
    # and FRHOT_BENCH_BACKEND=gloo exchanges through gloo (RCCL refuses two ranks on one device):
    # the N > 1 data path, its row bases and its merge on real kernels with one GPU.
    share_gpu = bool(os.environ.get("FRHOT_BENCH_SHARE_GPU"))
    backend = os.environ.get("FRHOT_BENCH_BACKEND", "nccl")          # "nccl" is RCCL on ROCm
    ndev = torch.cuda.device_count()                                  # does not initialise the GPU
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU (MI355X); none is visible")
    dev_index = 0 if share_gpu else local_rank
    if dev_index >= ndev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} needs GPU {dev_index}, this node shows {ndev}")
    if os.environ.get("FRHOT_BENCH_WATCHDOG_S"):      # diagnosis: say where a hung run stands, then end it
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FRHOT_BENCH_WATCHDOG_S"]), exit=True)
    if os.environ.get("FRHOT_BENCH_TRACE_EVERY_S"):   # diagnosis: where every rank stands, again and again
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["FRHOT_BENCH_TRACE_EVERY_S"]), repeat=True)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # FRHOT_BENCH_REHEARSE_DIST=1 with one rank: a one-rank RCCL group, and the gathers and the merge of the
    # multi-rank step run anyway (what a one-GPU box can rehearse of the N > 1 path)
    rehearse = world == 1 and bool(os.environ.get("FRHOT_BENCH_REHEARSE_DIST"))
    if world > 1 or rehearse:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        log(f"init_process_group({backend}) world {world}")
        kw = {"device_id": dev} if backend == "nccl" else {}
        # a rank that never arrives ends the job with an error after this long instead of hanging it
        dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180), **kw)

    from frhot import synth
    from frhot.dist import ShardedSearch, shard_rows
    from frhot.engine import Context, Gallery
    from frhot.weights import pack_blob, synthetic_weights

    P, B, D, K = world, args.batch, args.dim, args.k
    row_base, rows_local = shard_rows(args.gallery_rows, P, rank)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        log("context + model")
        ctx = Context(device=dev_index, max_batch=B, stream=stream.cuda_stream)
        ctx.load_model(pack_blob(synthetic_weights(seed=args.seed & 0xFFFF)), input_scale=1.0 / 16.0)
        log("crops")
        images = synth.crops(seed=args.seed + 3 + rank, n=B, lo=112, hi=448)
        buf, offsets, hw, strides = synth.pack_crops(images)
        crops_dev = torch.from_numpy(buf).to(dev)
        log(f"gallery shard: rows [{row_base}, {row_base + rows_local}) x {D}")
        gal = Gallery(ctx, D, rows_local)
        gal.fill_synthetic(seed=args.seed + 1000 * (rank + 1), row0=0, n=rows_local, nkeys=0, dup_every=100)

        if args.scan_shape_only:
            n_, d_, *shape = SCAN_SHAPES[args.scan_shape_only]
            shape = tuple(shape[:4]) + (5,)
            if (n_, d_) == (rows_local, D):
                obj, = scan_only(ctx, Gallery, dev, n_, d_, 0, [shape], gal=gal)
            else:
                obj, = scan_only(ctx, Gallery, dev, n_, d_, args.seed + 79, [shape])
            gal.close()
            ctx.close()
            os.write(json_fd, (json.dumps(obj) + "\n").encode())
            os.close(json_fd)
            return

        q_local = torch.empty((B, D), dtype=torch.float16, device=dev)
        flags = Context.flags(gamma=True, flip=True)

        def search_local(q_all, s_out, r_out):
            gal.search_dev(q_all.data_ptr(), None, q_all.shape[0], K, s_out.data_ptr(), r_out.data_ptr(), row_base)

        def merge(s_all, r_all, out_s, out_r):
            # the gathered blocks are merged where the all-gather left them
            ctx.merge_topk_packed_dev(r_all.data_ptr(), exchange.block_bytes, P, P * B, K, out_s.data_ptr(),
                                      out_r.data_ptr())

        exchange = ShardedSearch(P, B, D, K, dev, search_local, merge, always_exchange=rehearse,
                                 drain_before_exchange=share_gpu)
        sharded = exchange
        # FRHOT_BENCH_NATIVE_EXCHANGE=1: the same step through frhot_search_sharded_dev, the exchange inside
        # the library over its own RCCL communicator (the id travels over the torch group).  Opt-in: the
        # default line stays on the path the multi-GPU runs of earlier rounds measured.
        native = (P > 1 or rehearse) and bool(os.environ.get("FRHOT_BENCH_NATIVE_EXCHANGE")) and not share_gpu
        if native:
            box = [Context.dist_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ctx.dist_init(rank, P, box[0])
            nat_s = torch.empty((P * B, K), dtype=torch.float32, device=dev)
            nat_r = torch.empty((P * B, K), dtype=torch.int64, device=dev)

            def sharded(q):
                gal.search_sharded_dev(q.data_ptr(), None, B, K, nat_s.data_ptr(), nat_r.data_ptr(), row_base)
                return nat_s, nat_r
        last = {}

        def step():
            ctx.embed_dev(crops_dev.data_ptr(), offsets, hw, strides, flags, None, q_local.data_ptr(), D)
            last["s"], last["r"] = sharded(q_local)

        log("warmup")
        for _ in range(args.warmup):
            step()
        ctx.timing(True)
        if P > 1 or rehearse:
            dist.barrier()
        torch.cuda.synchronize()
        log("timed region")
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

        # Gate of the exchange path, on every rank (also with one): each rank asks for B rows that
        # are stored in ANOTHER rank's shard (rank r asks for rows of shard r+1), so a correct
        # answer needs the query gather, that shard's row base and the merge: top-1 must be the
        # asked global row with score 1 (its planted duplicate second, lower row first).
        log("exchange gate")
        pick = (np.arange(B, dtype=np.int64) * 7919 + 13) % max(rows_local, 1)
        pick[:4] = [98 % rows_local, 99 % rows_local, rows_local - 1, 0]
        mine16, _ = gal.read(pick)
        mine = torch.from_numpy(mine16).to(dev)
        if P > 1 or rehearse:
            every = torch.empty((P * B, D), dtype=torch.float16, device=dev)
            dist.all_gather_into_tensor(every, mine)
        else:
            every = mine
        src = (rank + 1) % P
        gs, gr = sharded(every[src * B:(src + 1) * B].contiguous())
        torch.cuda.synchronize()
        gs, gr = gs.cpu().numpy(), gr.cpu().numpy()
        native_same = None
        if native:          # the library's exchange against torch.distributed's on the same queries, bit for bit
            ts, tr = exchange(every[src * B:(src + 1) * B].contiguous())
            torch.cuda.synchronize()
            native_same = bool(np.array_equal(tr.cpu().numpy(), gr) and np.array_equal(ts.cpu().numpy(), gs))
        gate_ok = True
        for r in range(P):                      # the answer holds every rank's queries: check them all
            asked_rank = (r + 1) % P
            base_a, rows_a = shard_rows(args.gallery_rows, P, asked_rank)
            loc = (np.arange(B, dtype=np.int64) * 7919 + 13) % max(rows_a, 1)
            loc[:4] = [98 % rows_a, 99 % rows_a, rows_a - 1, 0]
            first = np.where(loc % 100 == 99, loc - 1, loc) + base_a          # a duplicate pair: lower row first
            second = np.where(loc % 100 == 98, loc + 1, np.where(loc % 100 == 99, loc, -1))
            second = np.where((second >= 0) & (second < rows_a), second + base_a, -1)
            blk_r, blk_s = gr[r * B:(r + 1) * B], gs[r * B:(r + 1) * B]
            gate_ok &= bool((blk_r[:, 0] == first).all() and (np.abs(blk_s[:, 0] - 1.0) <= 2e-3).all())
            has2 = second >= 0
            gate_ok &= bool((blk_r[has2, 1] == second[has2]).all())
        if P > 1 or rehearse:
            t = torch.tensor([1 if gate_ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            gate_ok = bool(int(t.item()))

        # PCIe-inclusive leg (one rank): the same crops handed over as pageable host arrays through
        # frhot_recognise (pack + upload + embed + search + answers back), host wall clock.
        host_leg = None
        if P == 1 and not rehearse and not args.no_host_leg:
            log("host-buffer leg")
            from concurrent.futures import ThreadPoolExecutor
            for _ in range(2):
                ctx.recognise(gal, images, None, K, gamma=True)
            # (enough batches that the first one's upload, which nothing hides, and the last one's lone
            # compute weigh little: 40 at the default 20 steps)
            n_it = 2 * max(2, min(2 * args.steps, 40) // 2)
            # two callers in flight, as the reference's handlers arrive from worker threads
            # (face_service.py:231-233): while one batch computes, the other's crops are packed and uploaded
            with ThreadPoolExecutor(max_workers=2) as pool:
                t0 = time.perf_counter()
                answers = list(pool.map(lambda _i: ctx.recognise(gal, images, None, K, gamma=True), range(n_it)))
                dt = time.perf_counter() - t0
            t1 = time.perf_counter()
            n_serial = max(3, min(args.steps // 2, 10))
            for _ in range(n_serial):
                ctx.recognise(gal, images, None, K, gamma=True)
            dt_serial = (time.perf_counter() - t1) / n_serial
            same = all(np.array_equal(a[1], res_r) for a in answers)
            host_leg = {"value": B * n_it / dt, "unit": "faces/s", "ms_per_batch": dt / n_it * 1e3, "batches": n_it,
                        "callers_in_flight": 2, "ms_per_batch_one_caller": dt_serial * 1e3,
                        "host_bytes_per_batch": int(sum(im.nbytes for im in images)),
                        "same_answer_as_device_resident": bool(same),
                        "what": "frhot_recognise from pageable host crops, answers back on the host"}

        # Secondary measurements on rank 0: scan-only shapes on galleries of their own.
        scan64 = scan4096 = scan4096_q64 = keyed = scan_rank8 = scan_10m_q64 = None
        if P == 1 and not args.no_scan64:      # single-GPU measurements: not repeated in the scaling runs
            log("scan-only shapes")
            n_, d_, *shape = SCAN_SHAPES["config2"]
            scan64, = scan_only(ctx, Gallery, dev, n_, d_, args.seed + 77, [tuple(shape)])
            if not args.no_scan_extra:
                n_, d_, *shape = SCAN_SHAPES["rank8"]
                scan_rank8, = scan_only(ctx, Gallery, dev, n_, d_, args.seed + 79, [tuple(shape)])
            if not args.no_scan_extra and (args.gallery_rows, D) == SCAN_SHAPES["10m_q64"][:2]:      # on the step's own resident gallery
                scan_10m_q64, = scan_only(ctx, Gallery, dev, rows_local, D, 0, [tuple(SCAN_SHAPES["10m_q64"][2:])], gal=gal)
            if not args.no_scan4096:
                scan4096, scan4096_q64, keyed = scan_only(
                    ctx, Gallery, dev, 1_000_000, 4096, args.seed + 78,
                    [(256, 5, "BASELINE config 3 at the reference's dimension (ridge)", "hbm", 10),
                     (64, 5, "the reference's dimension at a serving-sized batch of queries (HBM-bound)", "hbm", 10)],
                    keyed_run=500)

        # The check-in blur gate's measure over the same resident crops (not part of the timed step:
        # the reference applies it to check-in requests only, face_service.py:188-219).
        blur_leg = blur_sums = None
        if P == 1 and not args.no_scan64:
            log("blur gate")
            sums_dev = torch.empty((B, 2), dtype=torch.int64, device=dev)
            n_it = 20
            for _ in range(3):
                ctx.crop_sharpness_dev(crops_dev.data_ptr(), offsets, hw, strides, sums_dev.data_ptr())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(n_it):
                ctx.crop_sharpness_dev(crops_dev.data_ptr(), offsets, hw, strides, sums_dev.data_ptr())
            e1.record(stream)
            e1.synchronize()
            us = e0.elapsed_time(e1) / n_it * 1e3
            crop_bytes = int((hw[:, 0].astype(np.int64) * hw[:, 1] * 3).sum())
            blur_sums = sums_dev.cpu().numpy()
            blur_leg = {"kernel": "crop_sharpness_kernel", "bound": "hbm", "us_per_batch": us, "crops": B,
                        "algorithmic_bytes_per_launch": crop_bytes, "achieved": crop_bytes / us / 1e3,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": crop_bytes / us / 1e3 / HBM_PEAK_GBS,
                        "what": "Laplacian-variance sums of the batch's crops (memset + kernel), crops resident"}

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
                           "parallelism": f"dp{P}+rowshard{P}",
                           "exchange": "frhot_search_sharded_dev (RCCL inside the library)" if native
                           else "torch.distributed all-gathers + frhot_merge_topk_packed_dev"},
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
                "gates": {"exchange_top1_is_the_asked_row": gate_ok},
            }
            if native_same is not None:
                line["gates"]["library_exchange_equals_torch_exchange"] = native_same
            if host_leg is not None:
                line["value_pcie_inclusive"] = host_leg["value"]
                line["pcie_inclusive"] = host_leg
            if scan64 is not None:
                line["roofline_scan_hbm"] = scan64
                if scan_rank8 is not None:
                    line["roofline_scan_rank8"] = scan_rank8
                if scan_10m_q64 is not None:
                    line["roofline_scan_10m_q64"] = scan_10m_q64
                if scan4096 is not None:
                    line["roofline_scan_d4096"] = scan4096
                    line["roofline_scan_d4096_q64"] = scan4096_q64
                    line["search_inside_collections"] = keyed
            if blur_leg is not None:
                line["blur_gate"] = blur_leg
            if P == 1 and not rehearse and not args.no_detector:
                log("detector")
                line["detector"] = detector_leg(ctx, args)
            if P == 1 and not rehearse and not args.no_cpu_baseline:
                log("cpu baseline")
                base, gates = cpu_baseline(ctx, gal, images, args, rows_local, blur_sums)
                line["cpu_baseline"] = base
                line["gates"].update(gates)
        gal.close()
        ctx.close()
    if world > 1 or rehearse:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    if line is not None:
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    os.close(json_fd)
    if line is not None and not all(bool(v) for k, v in line["gates"].items() if isinstance(v, bool)):
        raise SystemExit("bench.py: a correctness gate failed: " + json.dumps(line["gates"]))


if __name__ == "__main__":
    main()
