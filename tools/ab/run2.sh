#!/bin/bash
# A/B of prebuilt libraries in ONE gpurun call, one bench mode at a time: tools/ab/run2.sh "<bench args>" lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
args="$1"; shift
for rep in 1 2; do for f in "$@"; do
  S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 200 $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-20s %-34s %.1f Gbp/s step %.3f ms tile %.3f km %.3f verified=%s' % ('$f'.split('/')[-1], '$args', d['value'], d['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms'], bool(d['verified_vs_oracle'])))" || echo "FAILED $f $args"
done; done
