#!/bin/bash
# one- and two-context step time against the size of the LAST chunk (S2K_DESC_TAIL = its fraction of the tiles): tools/ab/tail_taper.sh lib.so "<fractions>" "<chunks>"
cd $GRAFT_REPO_ROOT
f=$1
for c in $3; do for tl in $2; do
  S2K_DESC_CHUNKS=$c S2K_DESC_TAIL=$tl S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('chunks=%-3s tail=%-5s two-ctx %.3f ms  one-ctx %.3f  tile %.3f  exposed %.3f' % ('$c', '$tl', d['ms_per_step'], d['one_context']['ms_per_step'], r['kernel_ms'], r['kminmer_exposed_ms']))" || echo "FAILED $c $tl"
done; done
