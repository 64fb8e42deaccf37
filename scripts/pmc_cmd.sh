#!/bin/bash
# SQ counter pass over any command: scripts/pmc_cmd.sh <tag> <program> [args...]   (the program itself, e.g. python3 x.py or ./scripts/kbench fuse 128)
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/$1
shift
mkdir -p $out
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/sq -- "$@" > $out/sq.log 2>&1 || { tail -20 $out/sq.log; exit 1; }
python3 - "$out" <<'PY'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+"/sq/*/*counter_collection.csv")[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r["Kernel_Name"]][r["Counter_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
for k,c in agg.items():
    if "fill" in k or "rocclr" in k or "at::" in k: continue
    n=len(next(iter(c.values())))
    dur=sum(d for _,d in c["SQ_WAVE_CYCLES"])/n
    row={name: sum(v for v,_ in vals)/n for name,vals in c.items()}
    gui=row.get("GRBM_GUI_ACTIVE",0)
    clk=gui/8/dur if dur else 0   # GHz (cycles per ns)
    print(k[:100])
    print("   calls %d dur_us %.1f clk_GHz %.2f mfma_busy_frac %.3f  lds_active_frac %.3f bank_conf_frac %.3f wait_any/wave %.3f wait_inst_any/wave %.3f wait_lds/wave %.3f" % (
        n, dur/1e3, clk,
        row["SQ_VALU_MFMA_BUSY_CYCLES"]/ (gui/8*256*4) if gui else 0,
        row["SQ_LDS_IDX_ACTIVE"]/(gui/8*256) if gui else 0,
        row["SQ_LDS_BANK_CONFLICT"]/max(row["SQ_LDS_IDX_ACTIVE"],1),
        row["SQ_WAIT_ANY"]/max(row["SQ_WAVE_CYCLES"],1), row["SQ_WAIT_INST_ANY"]/max(row["SQ_WAVE_CYCLES"],1), row["SQ_WAIT_INST_LDS"]/max(row["SQ_WAVE_CYCLES"],1)))
PY
rm -rf $out/sq
