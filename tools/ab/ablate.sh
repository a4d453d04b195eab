#!/bin/bash
# tools/ab/ablate.sh <lib.so> "<bench args>" skip1 skip2 ...: tile-kernel time of a KNOBS build with parts of the kernel ablated
# (S2K_DEBUG_SKIP bits: 1 hash loop, 4 compaction, 128 listing + re-derivation + rounds, 64 re-derivation, 16 stores), one chunk per call.
cd $GRAFT_REPO_ROOT
lib=$1; args=$2; shift; shift
for s in "$@"; do
  S2K_LIB=$GRAFT_REPO_ROOT/$lib S2K_DEBUG_SKIP=$s S2K_DESC_CHUNKS=1 timeout -k 10 200 python bench.py --contexts 1 --steps 6 --warmup 2 --no-cpu-baseline --no-end-to-end --no-other-mode --verify-reads 0 $args 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s %-16s skip=%-4s step %.3f ms tile %.3f km %.3f' % ('$lib'.split('/')[-1], '$args', '$s', d['ms_per_step'], r['kernel_ms'], r['kminmer_kernel_ms']))" || echo "FAILED $lib $args skip=$s"
done
