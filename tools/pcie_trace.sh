#!/bin/bash
# phase times of s2k_extract (H2D / kernels / host allocation / D2H): KNOBS build of the host half only, on the GPU box
cd $GRAFT_REPO_ROOT
rm -f rust-seq2kminmers_amd/csrc/*.o && make -s -C rust-seq2kminmers_amd/csrc KNOBS=1 -j16 libs2k.so > /dev/null 2>&1 || exit 1
S2K_TRACE_EXTRACT=1 python tools/pcie_rate.py 2>&1 | grep -v amdgpu.ids
