#!/bin/bash
# tools/ab/hifi_ab.sh a.so b.so ...: the HiFi-like workload (BASELINE configs[3], sparse tiles in Hpc mode), step times per library, verified against the oracle sample
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for f in "$@"; do
  S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 300 python bench.py --workload hifi --steps 8 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 200 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-12s hifi two-ctx %.3f ms (%.1f Gbp/s)  one-ctx %.3f (%.1f)  tile %.3f verified=%s' % ('$f'.split('/')[-1], d['ms_per_step'], d['value'], d['one_context']['ms_per_step'], d['one_context']['value'], r['kernel_ms'], d['verified_vs_oracle']))" || echo "FAILED $f"
done; done
