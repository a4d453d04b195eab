#!/bin/bash
# instruction-cache behaviour of the tiled kernels (the unrolled hash loop alone is ~14 KB of code per instantiation)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_icache
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --contexts 1 --steps 2 --warmup 1 --no-cpu-baseline --verify-reads 0 > $out/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "hpc" if "tile_minimizer_kernel<31, true>" in n else "reg" if "tile_minimizer_kernel<31, false>" in n else "km" if "kminmer_kernel_fast" in n else None
        if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in agg:
    print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(agg[k].items())})
PY
