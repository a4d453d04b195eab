#!/bin/bash
# round 6: minimizer triples beside the k-min-mers (the k-min-mer kernel's non-FULL instantiation), round 5's library against HEAD's
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for f in tools/ab/base6.so rust-seq2kminmers_amd/csrc/libs2k.so; do
  S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-12s value %.1f  minimizers_only %s  other_mode %.1f  standalone_hpc %s' % ('$f'.split('/')[-1], d['value'], d['minimizers_only'], d['other_mode']['value'], d['standalone_hpc']['hpc']))" || echo FAILED $f
done; done
