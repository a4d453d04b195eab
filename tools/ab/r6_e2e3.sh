#!/bin/bash
# round 6: PCIe-inclusive rates, five repetitions of tools/pcie_rate.py in fresh processes (the host's run-to-run spread is large)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4 5; do echo "== run $rep"; timeout -k 10 300 python tools/pcie_rate.py 2>&1 | grep -E "Gbp/s"; done
