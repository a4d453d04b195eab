#!/bin/bash
# kernel times of the standalone hpc pipeline (two passes): rocprofv3 --kernel-trace --stats of tools/hpc_rate.py
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/hpcprof
rm -rf $out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/hpc_rate.py > $out.log 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("$out/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %4s avg %10.1f us  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
tail -1 $out.log
