#!/bin/bash
# round 6: PCIe-inclusive rates (s2k_extract host -> host, s2k_run_file) with the copy threads packing by AVX2 + pext (S2K_PACK_ISA=1, rounds 2-5) and by AVX-512 (default)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for isa in 1 2; do
  echo "== S2K_PACK_ISA=$isa"; S2K_PACK_ISA=$isa timeout -k 10 300 python tools/pcie_rate.py 2>&1 | grep -E "Gbp/s|rate|GB/s"
done; done
