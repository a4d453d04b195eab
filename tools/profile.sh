#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile.sh <tag>
# 1. rocprofv3 --kernel-trace --stats of the default bench command  -> gpurun_out/prof_<tag>/stats
# 2. separate PMC passes (FETCH_SIZE / WRITE_SIZE / SQ) on the same workload -> gpurun_out/prof_<tag>/pmc
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify-reads 0 > $out/stats.log 2>&1 || echo "stats pass failed"
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc/p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --verify-reads 0 > $out/pmc_p$i.log 2>&1 || echo "pmc pass $i failed"
done
python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/pmc > $out/pmc_summary.txt 2>&1
python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$out/stats/*/*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
with open("$out/kernel_stats_summary.csv", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify-reads 0\n")
    o.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "")
        n = n.split("(")[0][-60:]
        o.write('"%s",%s,%s,%s,%s\n' % (n, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]))
print(open("$out/kernel_stats_summary.csv").read())
PY
cat $out/pmc_summary.txt
python3 - <<PY
# HBM traffic of the headline (Hpc) minimizer kernel per launch, corrected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE reads exactly half of a wide coalesced read stream on gfx950 -> x2; WRITE_SIZE is exact; unit KB.
import json, re
txt = open("$out/pmc_summary.txt").read()
blk = txt.split("== tile_min<hpc>")[1].split("==")[0]
f = float(re.search(r"FETCH_SIZE\s+(\S+)", blk).group(1)); w = float(re.search(r"WRITE_SIZE\s+(\S+)", blk).group(1))
blk2 = txt.split("== tile_min<reg>")[1].split("==")[0] if "== tile_min<reg>" in txt else ""
out = {"mode": "hpc", "n_bases": 10000000000, "hbm_bytes_per_launch": int((2 * f + w) * 1024), "fetch_size_kb": f, "write_size_kb": w,
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on python3 bench.py --steps 2 --warmup 1; FETCH_SIZE x2 (gfx950)"}
if blk2:
    f2 = float(re.search(r"FETCH_SIZE\s+(\S+)", blk2).group(1)); w2 = float(re.search(r"WRITE_SIZE\s+(\S+)", blk2).group(1))
    out["regular_hbm_bytes_per_launch"] = int((2 * f2 + w2) * 1024)
json.dump(out, open("$out/traffic.json", "w"), indent=1)
print(out)
PY
