#!/bin/bash
# usage (on the GPU box): tools/variants_flags.sh "<extra hipcc flags 1>" "<extra flags 2>" ...   -- rebuilds the tiled kernels with them
cd $GRAFT_REPO_ROOT
run() {
  rm -f rust-seq2kminmers_amd/csrc/s2k_tile*.o
  make -s -C rust-seq2kminmers_amd/csrc -j16 libs2k.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $1" > /tmp/make.log 2>&1 || { echo "BUILD FAILED: $1"; tail -3 /tmp/make.log; return; }
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --verify-reads 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-70s hpc %.3f ms  reg %.3f ms  verified=%s' % ('$1', d['roofline']['kernel_ms'], d['other_mode']['kernel_ms'], d['verified_vs_oracle']))"
}
run ""
for f in "$@"; do run "$f"; done
