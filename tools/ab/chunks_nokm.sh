#!/bin/bash
# tools/ab/chunks_nokm.sh <knobs-lib.so> "<bench args>": what do the chunk boundaries of the tiled kernel cost by themselves?
# tile-kernel span with 1 / 3 / 6 / 8 / 12 chunks, with and without the k-min-mer stage on the second stream (S2K_DEBUG_NOKM, KNOBS build)
cd $GRAFT_REPO_ROOT
lib=$1; args=$2
for nokm in "" 1; do for ch in 1 3 6 8 12; do
  env S2K_LIB=$GRAFT_REPO_ROOT/$lib S2K_DESC_CHUNKS=$ch ${nokm:+S2K_DEBUG_NOKM=1} timeout -k 10 200 python bench.py --contexts 1 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 0 $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('chunks=%-2s km=%-3s step %.3f ms tile-span %.3f km-span %.3f' % ('$ch', 'off' if '$nokm' else 'on', d['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms']))" || echo "FAILED chunks=$ch nokm=$nokm"
done; done
