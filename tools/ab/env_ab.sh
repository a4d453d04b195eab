#!/bin/bash
# tools/ab/env_ab.sh <lib.so> "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ... : same library, different environments (two passes)
cd $GRAFT_REPO_ROOT
lib=$1; args=$2; shift; shift
for rep in 1 2; do for e in "$@"; do
  env S2K_LIB=$GRAFT_REPO_ROOT/$lib $e timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 200 $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-34s %-16s %.1f Gbp/s step %.3f ms tile %.3f km %.3f verified=%s' % ('$e', '$args', d['value'], d['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms'], bool(d['verified_vs_oracle'])))" || echo "FAILED $e $args"
done; done
