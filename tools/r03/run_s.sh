#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3/exp_feather_r3.log
: > $O
for fl in 16 0 16 0; do
  echo "== SQ_EXT_FLAGS=$fl ($([ $fl = 16 ] && echo 'groups of consecutive planes, round 2' || echo 'groups dealt, shipped'))" >> $O
  SQ_EXT_PLANES=10 SQ_EXT_FEATHER_ONLY=1 SQ_EXT_FLAGS=$fl timeout -k 10 300 python tools/ext_probe.py 2>&1 | grep feather >> $O
done
cat $O
