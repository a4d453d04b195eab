#!/bin/bash
# steadier A/B than run.sh: one chunk per call (the minimizer kernel alone on the caller's stream, the k-min-mer stage behind it), 30 steps, three alternations:
# tools/ab/steady.sh "<bench args>" a.so b.so ...
cd $GRAFT_REPO_ROOT
args="$1"; shift
for rep in ${REPS:-1 2 3}; do for f in "$@"; do
  S2K_DESC_CHUNKS=1 S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 200 python bench.py --contexts 1 --steps 30 --warmup 5 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 0 $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s %-14s one launch of the minimizer kernel %.3f ms  (step %.3f)' % ('$f'.split('/')[-1], '$args', r['kernel_ms'], d['ms_per_step']))" || echo "FAILED $f"
done; done
