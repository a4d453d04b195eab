#!/bin/bash
# tools/ab/pmc_km.sh "<bench args>" lib1.so ...: per-tile dynamic instruction counters of the k-min-mer kernel (desc_kminmer_kernel), two PMC passes, kernel-trace only, one chunk per call
cd /tmp && export TMPDIR=/tmp
args="$1"; shift
for f in "$@"; do
  for p in 1 2; do
    out=$GRAFT_REPO_ROOT/gpurun_out/pmckm_$(basename $f .so)_p$p
    rm -rf $out
    if [ $p = 1 ]; then set="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; else set="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES SQ_BUSY_CYCLES"; fi
    S2K_LIB=$GRAFT_REPO_ROOT/$f S2K_DESC_CHUNKS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 0 $args > $out.log 2>&1 || echo "failed $f"
  done
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list); dur = []
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmckm_$(basename $f .so)_p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "desc_kminmer_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmckm_$(basename $f .so)_p*/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "desc_kminmer_kernel" in r["Kernel_Name"]: dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
v = {c: sum(x) / len(x) / 1085070 for c, x in agg.items()}
print("%-12s %-16s" % ("$(basename $f)", "$args") + " ".join("%s=%.1f" % (c.replace("SQ_", ""), v[c]) for c in sorted(v)), "(per tile)  kernel %.3f ms (profiled, %d launches)" % (sum(dur) / max(len(dur), 1), len(dur)))
PY
done
