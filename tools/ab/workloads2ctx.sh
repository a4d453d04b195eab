#!/bin/bash
# tools/ab/workloads2ctx.sh "<chunk counts>" lib.so: two-context / one-context step times of the three workload shapes by chunks per call (S2K_DESC_CHUNKS; "" = the library's defaults)
cd $GRAFT_REPO_ROOT
cs="$1"; f=$2
for wl in c2 ont hifi; do for c in $cs; do
  if [ "$c" = "d" ]; then unset S2K_DESC_CHUNKS; else export S2K_DESC_CHUNKS=$c; fi
  S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 300 python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --no-other-configs --verify-reads 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-6s chunks=%-3s two-ctx %.3f ms (%.1f Gbp/s)  one-ctx %.3f (%.1f)  tile %.3f' % ('$wl', '$c', d['ms_per_step'], d['value'], d['one_context']['ms_per_step'], d['one_context']['value'], r['kernel_ms']))" || echo "FAILED $wl $c"
done; done
