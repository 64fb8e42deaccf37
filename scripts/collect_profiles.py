#!/usr/bin/env python3
"""Turn one GPU pass (scripts/gpu_round.sh <tag>, outputs under gpurun_out/<tag>) into the files
kept under profiles/: bench line, kernel-trace statistics, mean FETCH_SIZE / WRITE_SIZE per kernel
and the per-launch HBM traffic of the two scan workloads that bench.py reports.

Usage: python scripts/collect_profiles.py <tag> <run-name>      e.g.  r1e r01e
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys


def mean_counter(path, name):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}


def main():
    tag, run = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(root, "gpurun_out", tag)
    dst = os.path.join(root, "profiles")
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{run}_bench.json"))
    stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))[0]
    shutil.copy(stats, os.path.join(dst, f"{run}_bench_kernel_stats.csv"))
    fetch = mean_counter(glob.glob(os.path.join(src, "pmc_fetch", "*", "*counter_collection.csv"))[0], "FETCH_SIZE")
    write = mean_counter(glob.glob(os.path.join(src, "pmc_write", "*", "*counter_collection.csv"))[0], "WRITE_SIZE")
    out_csv = os.path.join(dst, f"{run}_bench_pmc_fetch_write.csv")
    with open(out_csv, "w") as f:
        f.write("Kernel_Name,dispatches,FETCH_SIZE_KiB_mean,WRITE_SIZE_KiB_mean\n")
        for k in sorted(fetch):
            if "rocclr" in k:
                continue
            f.write(f"\"{k}\",{fetch[k][0]},{fetch[k][1]:.0f},{write.get(k, (0, 0.0))[1]:.0f}\n")
    bench = json.load(open(os.path.join(src, "bench.json")))
    cfg = bench["config"]
    wanted = {
        "scan_topk_kernel<5,256>": ("scan_topk_kernelILi5ELi256E", {"rows": cfg["gallery_rows"], "dim": cfg["dim"],
                                                                     "queries": cfg["batch_per_gpu"], "k": cfg["k"]}),
        # (the 64-query tile with the selector wave; no key filter, whole-gallery form)
        "scan_sel_kernel<10>": ("scan_sel_kernelILi10ELb0ELb0E", {"rows": 1_000_000, "dim": cfg["dim"],
                                                                  "queries": 64, "k": 10}),
    }
    doc = {"run": run, "source": f"profiles/{run}_bench_pmc_fetch_write.csv",
           "correction": "FETCH_SIZE x2 (gfx950 counts 64-B units as 32 B), WRITE_SIZE as is; KiB x 1024", "kernels": {}}
    for label, (needle, workload) in wanted.items():
        hit = [k for k in fetch if needle in k]
        if not hit:
            continue
        fk, wk = fetch[hit[0]][1], write.get(hit[0], (0, 0.0))[1]
        doc["kernels"][label] = {"workload": workload, "fetch_kib": round(fk), "write_kib": round(wk),
                                 "traffic_bytes": int(round((2 * fk + wk) * 1024))}
    # scan-only shapes measured in counter passes of their own (bench.py --scan-shape-only)
    extra = {
        "rank8": ("scan_topk_kernel<5,256> at one rank's 8-GPU shape", "scan_topk_kernelILi5ELi256E",
                  {"rows": 1_250_000, "dim": 512, "queries": 2048, "k": 5}),
        "10m_q64": ("scan_sel_kernel<5> on the headline gallery", "scan_sel_kernelILi5ELb0ELb0E",
                    {"rows": 10_000_000, "dim": 512, "queries": 64, "k": 5}),
    }
    for shape, (label, needle, workload) in extra.items():
        fdir = glob.glob(os.path.join(src, f"pmc_FETCH_{shape}", "*", "*counter_collection.csv"))
        wdir = glob.glob(os.path.join(src, f"pmc_WRITE_{shape}", "*", "*counter_collection.csv"))
        if not fdir or not wdir:
            continue
        f2, w2 = mean_counter(fdir[0], "FETCH_SIZE"), mean_counter(wdir[0], "WRITE_SIZE")
        hit = [k for k in f2 if needle in k]
        if not hit:
            continue
        fk, wk = f2[hit[0]][1], w2.get(hit[0], (0, 0.0))[1]
        doc["kernels"][label] = {"workload": workload, "fetch_kib": round(fk), "write_kib": round(wk),
                                 "traffic_bytes": int(round((2 * fk + wk) * 1024)),
                                 "source": f"separate rocprofv3 --pmc passes of bench.py --scan-shape-only {shape}"}
    long_run = os.path.join(src, "bench_2000_steps.json")
    if os.path.exists(long_run):
        shutil.copy(long_run, os.path.join(dst, f"{run}_bench_2000_steps.json"))
    with open(os.path.join(dst, f"{run}_pmc_traffic.json"), "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc["kernels"], indent=1))


if __name__ == "__main__":
    main()
