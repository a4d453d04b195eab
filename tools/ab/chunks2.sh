#!/bin/bash
# two-context step time against the number of chunks per call: tools/ab/chunks2.sh "<chunk counts>" a.so b.so ...
cd $GRAFT_REPO_ROOT
cs="$1"; shift
for f in "$@"; do for c in $cs; do
  S2K_DESC_CHUNKS=$c S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-16s chunks=%-3s two-ctx %.3f ms (%.1f Gbp/s)  one-ctx %.3f  tile %.3f  exposed %.3f' % ('$f'.split('/')[-1], '$c', d['ms_per_step'], d['value'], d['one_context']['ms_per_step'], r['kernel_ms'], r['kminmer_exposed_ms']))" || echo "FAILED $f $c"
done; done
