#!/bin/bash
# like tools/variants.sh, for csrc/s2k_kminmer.hip
cd $GRAFT_REPO_ROOT
H=rust-seq2kminmers_amd/csrc/s2k_kminmer.hip
cp $H /tmp/km.orig
run() {
  rm -f rust-seq2kminmers_amd/csrc/s2k_kminmer.o
  make -s -C rust-seq2kminmers_amd/csrc -j16 libs2k.so > /tmp/make.log 2>&1 || { echo "BUILD FAILED: $1"; tail -3 /tmp/make.log; return; }
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify-reads 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-70s km hpc %.3f reg %.3f ms  (%.1f / %.1f Gbp/s) verified=%s' % ('$1', d['roofline']['kminmer_kernel_ms'], d['other_mode']['kminmer_kernel_ms'], d['value'], d['other_mode']['value'], d['verified_vs_oracle']))"
}
run "baseline"
for e in "$@"; do
  cp /tmp/km.orig $H
  sed -i "$e" $H
  if cmp -s /tmp/km.orig $H; then echo "NO CHANGE: $e"; continue; fi
  run "$e"
done
cp /tmp/km.orig $H
