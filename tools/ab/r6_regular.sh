#!/bin/bash
# round 6: the Regular family's k-min-mer stage alone behind the minimizer kernel: staged (line-filling) stores against the lane-serial kernel's direct ones and round 5's kernel
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in "tools/ab/base6.so 0" "tools/ab/km2.so 0" "tools/ab/km2.so 1"; do set -- $v
  if [ "$2" = 1 ]; then export S2K_KM_NO_COAL=1; else unset S2K_KM_NO_COAL; fi
  S2K_LIB=$GRAFT_REPO_ROOT/$1 timeout -k 10 200 python bench.py --mode regular --steps 12 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 300 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s no_coal=%s regular value %.1f  one-ctx %.1f (%.3f ms)  tile %.3f  km-kernel %.3f  verified=%s' % ('$1'.split('/')[-1], '$2', d['value'], d['one_context']['value'], d['one_context']['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms'], bool(d['verified_vs_oracle'])))" || echo "FAILED $v"
done; done
