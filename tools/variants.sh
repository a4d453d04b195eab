#!/bin/bash
# usage (on the GPU box): tools/variants.sh "<sed expr 1>" "<sed expr 2>" ...
# Rebuilds the tiled kernel with each edit of csrc/s2k_tile_impl.h applied on its own (the box's tree is scratch) and prints
# the kernel times of the default bench.  First line = unmodified tree.
cd $GRAFT_REPO_ROOT
H=rust-seq2kminmers_amd/csrc/s2k_tile_impl.h
cp $H /tmp/impl.orig
trap 'cp /tmp/impl.orig $H' EXIT  # also when the script is killed half way
run() {
  rm -f rust-seq2kminmers_amd/csrc/s2k_tile*.o  # (object mtimes come from the build host: force the rebuild)
  make -s -C rust-seq2kminmers_amd/csrc -j16 libs2k.so > /tmp/make.log 2>&1 || { echo "BUILD FAILED: $1"; tail -3 /tmp/make.log; return; }
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-end-to-end --verify-reads 100 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-60s hpc %.3f ms  reg %.3f ms  (pipeline %.2f / %.2f Gbp/s) verified=%s' % ('$1', d['roofline']['kernel_ms'], d['other_mode']['kernel_ms'], d['value'], d['other_mode']['value'], d['verified_vs_oracle']))"
}
run "baseline"
for e in "$@"; do
  cp /tmp/impl.orig $H
  sed -i "$e" $H
  if cmp -s /tmp/impl.orig $H; then echo "NO CHANGE: $e"; continue; fi
  run "$e"
done
cp /tmp/impl.orig $H
