#!/bin/bash
# tools/isa/regs.sh "<extra -D flags>" [l]: compiles the tiled kernel for one compile-time l (default 31) to ISA in /tmp and prints the
# register use of its four instantiations (Hpc/Regular x descriptor/legacy) -- a CPU-only check before a variant goes to the GPU.
extra=$1; l=${2:-31}
cd $(dirname $0)/../../rust-seq2kminmers_amd/csrc
out=/tmp/isa/v_$(echo "$extra" | tr -c 'A-Za-z0-9=\n' '_')_L$l.s
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -mllvm -amdgpu-atomic-optimizer-strategy=None $extra -DS2K_TILE_L=$l --cuda-device-only -S -o $out s2k_tile_inst.hip || exit 1
awk '/\.name:.*tile_minimizer_kernel/{n=$2} /\.sgpr_spill_count/{ss=$2} /\.vgpr_count/{v=$2} /\.vgpr_spill_count/{vs=$2; if(n!=""){sub(/.*kernelILi/,"",n); sub(/EEEv.*/,"",n); print n, "vgpr", v, "vspill", vs, "sspill", ss; n=""}} /\.private_segment_fixed_size/{}' $out
echo $out
