#!/bin/bash
# tools/pmc_occ.sh: LDS / wait counters of the tiled kernel at 2 and 3 blocks per CU (S2K_DEBUG_BLOCKS_PER_CU)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_occ
mkdir -p $out
# the ablation knobs exist only in KNOBS builds (the copy of the tree on the GPU box is scratch)
rm -f $GRAFT_REPO_ROOT/rust-seq2kminmers_amd/csrc/*.o && make -s -C $GRAFT_REPO_ROOT/rust-seq2kminmers_amd/csrc KNOBS=1 -j8 libs2k.so > /dev/null 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
for b in 2 3; do
 i=0
 for set in "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_BUSY_CYCLES" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  S2K_DEBUG_BLOCKS_PER_CU=$b timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/b${b}_$i -- python3 $GRAFT_REPO_ROOT/bench.py --contexts 1 --steps 1 --warmup 1 --no-cpu-baseline --verify-reads 0 > $out/b${b}_$i.log 2>&1 || echo "b=$b set $i failed"
 done
done
python3 - <<PY
import csv, glob, collections
for b in (2, 3):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$out/b%d_*/*/*counter_collection.csv" % b):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = "hpc" if "tile_minimizer_kernel<31, true>" in n else "reg" if "tile_minimizer_kernel<31, false>" in n else None
            if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(agg):
        print("blocks/CU=%d %s " % (b, k) + " ".join("%s=%.4g" % (c.replace("SQ_",""), sum(v) / len(v) / 1085070) for c, v in sorted(agg[k].items())), "(per tile)")
PY
