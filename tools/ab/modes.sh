#!/bin/bash
# tools/ab/modes.sh "<modes>" lib1.so lib2.so ...: device-resident rates (tools/mode_rates.py) of prebuilt libraries, two passes
cd $GRAFT_REPO_ROOT
modes="$1"; shift
for rep in 1 2; do for f in "$@"; do
  S2K_LIB=$GRAFT_REPO_ROOT/$f timeout -k 10 200 python tools/mode_rates.py $modes 2>&1 | grep chunks | sed "s|^|$(basename $f) |"
done; done
