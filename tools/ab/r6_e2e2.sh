#!/bin/bash
# round 6: PCIe-inclusive rates with the result arrays pinned (direct DMA of the drain, default) and not (S2K_PIN_RESULTS=0: the staging ring + copy threads of rounds 2-5)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for pin in 0 1; do
  echo "== S2K_PIN_RESULTS=$pin"; S2K_PIN_RESULTS=$pin timeout -k 10 300 python tools/pcie_rate.py 2>&1 | grep -E "Gbp/s"
done; done
