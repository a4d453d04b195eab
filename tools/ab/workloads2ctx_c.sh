#!/bin/bash
# c2-shaped calls of other sizes: tools/ab/workloads2ctx_c.sh <reads> "<chunk counts>" lib.so
cd $GRAFT_REPO_ROOT
n=$1; cs="$2"; f=$3
for c in $cs; do
  if [ "$c" = "d" ]; then unset S2K_DESC_CHUNKS; else export S2K_DESC_CHUNKS=$c; fi
  S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 300 python bench.py --reads $n --steps 12 --warmup 4 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('c2 x %-8s chunks=%-3s two-ctx %.3f ms (%.1f Gbp/s)  one-ctx %.3f (%.1f)  tile %.3f' % ('$n', '$c', d['ms_per_step'], d['value'], d['one_context']['ms_per_step'], d['one_context']['value'], r['kernel_ms']))" || echo "FAILED $n $c"
done
