#!/bin/bash
# tools/ab/pmc2.sh "<bench args>" lib1.so lib2.so ...: dynamic instruction counts of the tiled kernel per tile for prebuilt variant libraries
# (rocprofv3 --pmc, kernel-trace only; one chunk per call so that one launch covers all tiles)
cd /tmp && export TMPDIR=/tmp
args="$1"; shift
for f in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc2_$(basename $f .so)
  rm -rf $out
  S2K_LIB=$GRAFT_REPO_ROOT/$f S2K_DESC_CHUNKS=1 timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 0 $args > $out.log 2>&1 || echo "failed $f"
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "tile_minimizer_kernel" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-14s %-16s" % ("$(basename $f)", "$args") + " ".join("%s=%.1f" % (c.replace("SQ_", ""), sum(v) / len(v) / 1085070) for c, v in sorted(agg.items())), "(per tile)")
PY
done
