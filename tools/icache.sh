#!/bin/bash
# usage (on the GPU box, from the repo root): tools/icache.sh <tag>
# Instruction-cache behaviour of the tiled kernel (65 KB of code for <31, Hpc>, 41 KB for <31, Regular>; the L1I is 64 KB per two
# CUs): rocprofv3 --pmc passes (kernel-trace only), one chunk per call so that a launch covers all tiles.
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/icache_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_TC_INST_REQ SQ_IFETCH SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES" "InstrFetchLatency" "LdsLatency"; do
  i=$((i+1))
  S2K_DESC_CHUNKS=1 timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline --verify-reads 0 > $out/p$i.log 2>&1 || echo "set $i failed"
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "hpc" if "tile_minimizer_kernel<31, true," in n else "reg" if "tile_minimizer_kernel<31, false," in n else "km" if "desc_kminmer" in n else None
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, " ".join("%s=%.5g" % (c, sum(v) / len(v)) for c, v in sorted(agg[k].items())), "(per launch)")
PY
