#!/bin/bash
# round 6: PCIe-inclusive rates with the copy threads bound to the GPU's NUMA node (default) and unbound (S2K_NUMA_BIND=0), fresh processes, alternating
cd $GRAFT_REPO_ROOT
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3
for rep in 1 2 3 4; do for b in 1 0; do echo "== run $rep S2K_NUMA_BIND=$b"; S2K_NUMA_BIND=$b S2K_PCIE_SKIP_FILE=$( [ $rep -gt 2 ] && echo 1 ) timeout -k 10 300 python tools/pcie_rate.py 2>&1 | grep -E "Gbp/s"; done; done
