#!/bin/bash
# timing-only A/B of prebuilt libraries whose results may be WRONG (ablations): tools/ab/time_only.sh "<bench args>" a.so b.so ...
cd $GRAFT_REPO_ROOT
args="$1"; shift
for rep in 1 2; do for f in "$@"; do
  S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 200 python bench.py --contexts 1 --steps 8 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 0 $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-20s %-16s step %.3f ms tile %.3f km-exposed %.3f' % ('$f'.split('/')[-1], '$args', d['ms_per_step'], r['kernel_ms'], r['kminmer_exposed_ms']))" || echo "FAILED $f"
done; done
