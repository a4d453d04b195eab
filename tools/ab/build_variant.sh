#!/bin/bash
# tools/ab/build_variant.sh <name> "<extra hipcc flags>": builds the library from the working tree in a scratch copy with extra
# -D flags and leaves it as tools/ab/<name>.so (git-ignored) for tools/ab/run2.sh
set -e
name=$1; extra=$2
root=$(cd $(dirname $0)/../.. && pwd)
tmp=$(mktemp -d /tmp/s2k_variant.XXXXXX)
mkdir -p $tmp/rust-seq2kminmers_amd $tmp/include
cp -r $root/rust-seq2kminmers_amd/csrc $tmp/rust-seq2kminmers_amd/csrc
cp $root/include/*.h $root/include/*.hpp $tmp/include/
rm -f $tmp/rust-seq2kminmers_amd/csrc/*.o $tmp/rust-seq2kminmers_amd/csrc/*.so
make -s -C $tmp/rust-seq2kminmers_amd/csrc -j8 libs2k.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra"
cp $tmp/rust-seq2kminmers_amd/csrc/libs2k.so $root/tools/ab/$name.so
rm -rf $tmp
echo built tools/ab/$name.so
