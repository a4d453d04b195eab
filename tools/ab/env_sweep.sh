#!/bin/bash
# one library, several environments: tools/ab/env_sweep.sh lib.so "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...
cd $GRAFT_REPO_ROOT
f=$1; args=$2; shift; shift
for rep in 1 2; do for e in "$@"; do
  env $e S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 100 $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-44s two-ctx %.3f ms (%.0f)  one-ctx %.3f (%.0f)  tile %.3f  km %.3f  exposed %.3f  ok=%s' % ('$e', d['ms_per_step'], d['value'], d['one_context']['ms_per_step'], d['one_context']['value'], r['kernel_ms'], r['kminmer_kernel_ms'], r['kminmer_exposed_ms'], d['verified_vs_oracle']['ok']))" || echo "FAILED $e"
done; done
