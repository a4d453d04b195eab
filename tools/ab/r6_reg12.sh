#!/bin/bash
# round 6: Regular family, 16-wave minimizer kernel with the k-min-mer stage behind it (default) against a 12-wave kernel with the k-min-mer stage beside it in chunks
cd $GRAFT_REPO_ROOT
run() { local label=$1 lib=$2; shift 2
  env "$@" S2K_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python bench.py --mode regular --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s value %.1f (%.3f ms)  one-ctx %.1f (%.3f ms)  tile %.3f  km-span %.3f  exposed %.3f  verified=%s' % ('$label', d['value'], d['ms_per_step'], d['one_context']['value'], d['one_context']['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms'], r['kminmer_exposed_ms'], bool(d['verified_vs_oracle'])))" || echo "FAILED $label"
}
for rep in 1 2; do
  run "km3 (16 waves, serial)" tools/ab/km3.so S2K_X=0
  for c in 3 4 6 8; do run "reg12 chunks=$c" tools/ab/reg12.so S2K_DESC_CHUNKS=$c; done
  run "km3 16 waves chunks=4" tools/ab/km3.so S2K_DESC_CHUNKS=4
done
