#!/bin/bash
# round 6: Regular family through the kernel without a tile buffer (S2K_STREAM_KERNEL=1) against the tiled one, k-min-mer stage behind it or beside it in chunks
cd $GRAFT_REPO_ROOT
run() { local label=$1 lib=$2; shift 2
  env "$@" S2K_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python bench.py --mode regular --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 300 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s value %.1f (%.3f ms)  one-ctx %.1f (%.3f ms)  tile %.3f  km-span %.3f  exposed %.3f  verified=%s' % ('$label', d['value'], d['ms_per_step'], d['one_context']['value'], d['one_context']['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms'], r['kminmer_exposed_ms'], bool(d['verified_vs_oracle'])))" || echo "FAILED $label"
}
for rep in 1 2; do
  run "tiled (16 waves, serial)" tools/ab/km7.so S2K_X=0
  run "stream 4w serial" tools/ab/km7.so S2K_STREAM_KERNEL=1
  run "stream 5w serial" tools/ab/st5.so S2K_STREAM_KERNEL=1
  for c in 4 6 8; do
    run "stream 4w(107) chunks=$c" tools/ab/km7.so S2K_STREAM_KERNEL=1 S2K_DESC_CHUNKS=$c
    run "stream 4w(96) chunks=$c" tools/ab/st4f.so S2K_STREAM_KERNEL=1 S2K_DESC_CHUNKS=$c
    run "stream 5w chunks=$c" tools/ab/st5.so S2K_STREAM_KERNEL=1 S2K_DESC_CHUNKS=$c
  done
done
