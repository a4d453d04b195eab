#!/bin/bash
# (prio1.so / prio3.so of this sweep were builds with -DS2K_TILE_PRIO=1 / 3 -- one s_setprio at the head of the tile kernel -- of commit "k-min-mer kernel: records fetched with the first round trip ..."; the experiment code has since been removed)
# round 6: Hpc step against the chunk count (S2K_DESC_CHUNKS), with and without staged stores for the last chunk's k-min-mer kernel (S2K_KM_TAIL_COAL), and the
# tile kernel's waves at a raised issue priority: tools/ab/r6_sweep.sh
cd $GRAFT_REPO_ROOT
run() { # label, lib, env...
  local label=$1 lib=$2; shift 2
  env "$@" S2K_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python bench.py --mode hpc --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s value %.1f (%.3f ms)  one-ctx %.1f (%.3f ms)  tile %.3f  km-span %.3f  exposed %.3f  verified=%s' % ('$label', d['value'], d['ms_per_step'], d['one_context']['value'], d['one_context']['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms'], r['kminmer_exposed_ms'], bool(d['verified_vs_oracle'])))" || echo "FAILED $label"
}
for rep in 1 2; do
  run "km3 default" tools/ab/km3.so S2K_X=0
  run "km3 tail-coal" tools/ab/km3.so S2K_KM_TAIL_COAL=1
  for c in 3 4 5 8; do run "km3 chunks=$c" tools/ab/km3.so S2K_DESC_CHUNKS=$c; run "km3 chunks=$c tail-coal" tools/ab/km3.so S2K_DESC_CHUNKS=$c S2K_KM_TAIL_COAL=1; done
  run "prio1" tools/ab/prio1.so S2K_X=0
  run "prio3" tools/ab/prio3.so S2K_X=0
done
