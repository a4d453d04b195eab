#!/bin/bash
# tools/experiments/build_hlb.sh <name> "<-D flags>": one variant of hash_loop_bench (binary tools/experiments/hlb_<name>, ISA /tmp/hx/<name>.s)
cd $(dirname $0)/../..
n=$1; f=$2; mkdir -p /tmp/hx
F="-O3 -std=c++17 --offload-arch=gfx950 -I rust-seq2kminmers_amd/csrc -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-result -DHX_TAG=\"$n\" $f"
/opt/rocm/bin/hipcc $F tools/experiments/hash_loop_bench.hip -o tools/experiments/hlb_$n 2>&1 | grep -A5 "error" 
/opt/rocm/bin/hipcc $F --cuda-device-only -S tools/experiments/hash_loop_bench.hip -o /tmp/hx/$n.s 2>&1 | grep -A5 "error"
grep -E "^\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count):" /tmp/hx/$n.s | paste - - - | sed "s/^/$n: /"
