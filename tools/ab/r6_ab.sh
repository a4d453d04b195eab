#!/bin/bash
# round 6: A/B of prebuilt libraries, Hpc and Regular, one context and the default two; tools/ab/r6_ab.sh a.so b.so ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for m in ${MODES:-hpc regular}; do for f in "$@"; do
  S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 200 python bench.py --mode $m --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 300 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s %-8s value %.1f  one-ctx %.1f (%.3f ms)  tile %.3f  km-kernel %.3f  exposed %.3f  verified=%s' % ('$f'.split('/')[-1], '$m', d['value'], d['one_context']['value'], d['one_context']['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms'], r['kminmer_exposed_ms'], bool(d['verified_vs_oracle'])))" || echo "FAILED $f $m"
done; done; done
