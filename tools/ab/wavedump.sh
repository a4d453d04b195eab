#!/bin/bash
# where did the waves of the tiled kernel run, and when did they finish?  tools/ab/wavedump.sh "<bench args>" lib1.so ... (KNOBS/PROFILE variant builds)
cd $GRAFT_REPO_ROOT
args="$1"; shift
for f in "$@"; do
  b=$(basename $f .so)
  S2K_LIB=$GRAFT_REPO_ROOT/$f S2K_DESC_CHUNKS=1 S2K_DEBUG_SKIP=32 S2K_DEBUG_WAVE_DUMP=$GRAFT_REPO_ROOT/gpurun_out/waves_$b.txt timeout -k 10 200 python bench.py --contexts 1 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 0 --steps 1 --warmup 0 $args 2>&1 | grep "dbg\|value" | cut -c1-300
  python3 - <<PY
import collections
rows = [tuple(map(int, l.split())) for l in open("$GRAFT_REPO_ROOT/gpurun_out/waves_$b.txt")]
print("$b: waves", len(rows))
# HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
cus = collections.Counter(); simds = collections.Counter()
for i, t, xcc, hw in rows:
    cu = (xcc & 0xF, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xF)
    cus[cu] += 1; simds[cu + ((hw >> 4) & 3,)] += 1
print(" distinct CUs", len(cus), "waves per CU", collections.Counter(cus.values()), "waves per SIMD", collections.Counter(simds.values()))
ts = sorted(r[1] for r in rows)
print(" finish times (10 ns ticks): min %d median %d max %d" % (ts[0], ts[len(ts)//2], ts[-1]))
PY
done
