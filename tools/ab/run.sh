#!/bin/bash
# A/B of prebuilt libraries in ONE gpurun call: tools/ab/run.sh tools/ab/a.so tools/ab/b.so ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for f in "$@"; do
  S2K_LIB=$GRAFT_REPO_ROOT/$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --verify-reads 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; o=d['other_mode']; print('%-24s hpc: %.1f Gbp/s step %.3f one-ctx %.3f tile %.3f km %.3f | reg: %.1f Gbp/s tile %.3f km %.3f pipe %.3f  verified=%s' % ('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['one_context']['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms'], o['value'], o['kernel_ms'], o['kminmer_kernel_ms'], o['pipeline_ms'], d['verified_vs_oracle']))"
done; done
