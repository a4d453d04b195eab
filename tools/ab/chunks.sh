#!/bin/bash
# chunks of tiles per call on the descriptor path (second stream for scan + k-min-mer kernel): tools/ab/chunks.sh lib.so "1 4 6 8"
cd $GRAFT_REPO_ROOT
lib=$1
for mode in hpc regular; do for n in $2; do
  S2K_DESC_CHUNKS=$n S2K_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-mode --verify-reads 2000 --mode $mode 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('chunks %-3s %-8s %.1f Gbp/s step %.3f ms (wall) kernels-span %.3f tile-span %.3f km-span %.3f verified=%s' % ('$n', '$mode', d['value'], d['ms_per_step'], r['time_ms'], r['kernel_ms'], r['kminmer_kernel_ms'], bool(d['verified_vs_oracle'])))" || echo "FAILED chunks=$n $mode"
done; done
