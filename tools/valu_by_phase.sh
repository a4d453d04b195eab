#!/bin/bash
# usage (on the GPU box, from the repo root): tools/valu_by_phase.sh <tag>
# Dynamic VALU/LDS instruction counts of the tiled kernel per phase, by ablation: the kernel is profiled
# (rocprofv3 --pmc, kernel-trace only) with S2K_DEBUG_SKIP = 0 / 1 (no hash loop) / 4 (no compaction) / 128 (dense phase stops after
# the tile's word: no listing, re-derivation, rounds) / 64 (no re-derivation) / 16 (no stores); one chunk per call so that a launch
# covers all tiles.  (2 = no dense phase at all is not safe on the descriptor path: the k-min-mer kernel would read tile words
# nobody wrote.)
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out/valu_$tag
mkdir -p $out
# the ablation knobs exist only in KNOBS builds (the copy of the tree on the GPU box is scratch)
rm -f $GRAFT_REPO_ROOT/rust-seq2kminmers_amd/csrc/*.o && make -s -C $GRAFT_REPO_ROOT/rust-seq2kminmers_amd/csrc KNOBS=1 -j8 libs2k.so > /dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
for skip in 0 1 4 128 64 16; do
  S2K_DESC_CHUNKS=1 S2K_DEBUG_SKIP=$skip timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $out/s$skip -- python3 $GRAFT_REPO_ROOT/bench.py --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline --verify-reads 0 > $out/s$skip.log 2>&1 || echo "skip=$skip failed"
done
python3 - <<PY
import csv, glob, collections
for skip in (0, 1, 4, 128, 64, 16):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$out/s%d/*/*counter_collection.csv" % skip):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = "hpc" if "tile_minimizer_kernel<31, true," in n else "reg" if "tile_minimizer_kernel<31, false," in n else None
            if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob("$out/s%d/*/*kernel_trace.csv" % skip):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = "hpc" if "tile_minimizer_kernel<31, true," in n else "reg" if "tile_minimizer_kernel<31, false," in n else None
            if k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k in sorted(agg):
        print("skip=%d %s kernel_ms=%.3f " % (skip, k, sum(dur[k]) / max(len(dur[k]), 1)) + " ".join("%s=%.4g" % (c, sum(v) / len(v) / 1085070) for c, v in sorted(agg[k].items())), "(per tile)")
PY
