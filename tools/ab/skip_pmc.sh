#!/bin/bash
# tools/ab/skip_pmc.sh knobs.so: dynamic instruction counts of the Hpc tile kernel per phase, by ablation (S2K_DEBUG_SKIP bits of a KNOBS build; results are wrong with a bit set,
# only the counters are read): 1 no hash loop, 2 no dense phase, 128 no listing / re-derivation / rounds, 64 no re-derivation, 16 no stores
for s in 0 1 2 128 64 16; do
  echo "== S2K_DEBUG_SKIP=$s"
  S2K_DEBUG_SKIP=$s $GRAFT_REPO_ROOT/tools/ab/pmc3.sh "--mode hpc" $1 2>&1 | tail -1
done
