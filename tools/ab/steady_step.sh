#!/bin/bash
# whole-step A/B with the library's default chunking, 30 steps, one and two contexts: tools/ab/steady_step.sh "<bench args>" a.so b.so ...
cd $GRAFT_REPO_ROOT
args="$1"; shift
for rep in ${REPS:-1 2 3}; do for f in "$@"; do
  S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 0 $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s %-14s two-ctx %.3f ms  one-ctx %.3f  (minimizer kernels %.3f, k-min-mer kernels %.3f)' % ('$f'.split('/')[-1], '$args', d['ms_per_step'], d['one_context']['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms']))" || echo "FAILED $f"
done; done
