#!/bin/bash
# Regular mode: three waves per SIMD + the k-min-mer kernel beside them (shipped) against four waves per SIMD (16-wave blocks, nothing fits beside them):
# one-context and two-context step time by the number of chunks.  tools/ab/regular_occupancy.sh head.so tw16.so
cd $GRAFT_REPO_ROOT
for f in "$@"; do for c in 1 2 4 8; do
  S2K_DESC_CHUNKS=$c S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 200 python bench.py --mode regular --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-10s chunks=%-2s two-ctx %.3f ms (%.0f Gbp/s)  one-ctx %.3f (%.0f)  minimizer kernels %.3f  k-min-mer span %.3f  exposed %.3f  verified=%s' % ('$f'.split('/')[-1], '$c', d['ms_per_step'], d['value'], d['one_context']['ms_per_step'], d['one_context']['value'], r['kernel_ms'], r['kminmer_kernel_ms'], r['kminmer_exposed_ms'], d['verified_vs_oracle']['ok']))" || echo "FAILED $f $c"
done; done
