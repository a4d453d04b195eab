#!/bin/bash
# standalone hpc, same-call A/B of prebuilt libraries: tools/ab/r6_hpc_ab.sh a.so b.so ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for f in "$@"; do
  S2K_LIB=$GRAFT_REPO_ROOT/$f python tools/hpc_rate.py 2>&1 | grep standalone | sed "s|^|$(basename $f) |"
done; done
for f in "$@"; do S2K_LIB=$GRAFT_REPO_ROOT/$f python tools/hpc_rate.py rle 2>&1 | grep standalone | sed "s|^|$(basename $f) |"; done
