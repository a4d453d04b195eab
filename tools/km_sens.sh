#!/bin/bash
# sensitivity of the k-min-mer kernel to its VALU work: window loop cut to one term (results wrong, timing only)
cd $GRAFT_REPO_ROOT
H=rust-seq2kminmers_amd/csrc/s2k_kminmer.hip
cp $H /tmp/km.orig
run() {
  rm -f rust-seq2kminmers_amd/csrc/s2k_kminmer.o
  make -s -C rust-seq2kminmers_amd/csrc -j16 libs2k.so > /tmp/make.log 2>&1 || { echo "BUILD FAILED: $1"; tail -3 /tmp/make.log; return; }
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify-reads 0 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-50s km hpc %.3f reg %.3f ms' % ('$1', d['roofline']['kminmer_kernel_ms'], d['other_mode']['kminmer_kernel_ms']))"
}
run baseline
sed -i '284s/m < k/m < 1/' $H; run "window loop: 1 term instead of k"
cp /tmp/km.orig $H
sed -i 's/if (o_hash) o_hash\[o\] = hmin;/if (o_hash \&\& o == 12345) o_hash[o] = hmin;/; s/if (o_start) o_start\[o\] = j\[u\];/if (o_start \&\& o == 12345) o_start[o] = j[u];/; s/if (o_end) o_end\[o\] = end;/if (o_end \&\& o == 12345) o_end[o] = end;/; s/if (o_rev) o_rev\[o\] = (uint8_t)(Rv < F);/if (o_rev \&\& o == 12345) o_rev[o] = (uint8_t)(Rv < F);/' $H; run "no output stores"
cp /tmp/km.orig $H
