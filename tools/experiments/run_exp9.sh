#!/bin/bash
# round-2 experiment 9: feather kernels of round 1 (tree at _r1/) against this round's on the same box; plane-group tests
set -o pipefail
mkdir -p gpurun_out
CS=image-stitcher_amd/csrc
L=$PWD/gpurun_out/r2_exp9.log
: > $L
timeout -k 10 900 python -m pytest tests/test_fuse_gpu.py -x -q -k "feather or fuzz or plane_groups" >> $L 2>&1; echo "pytest rc $?"; tail -3 $L
for rep in 1 2; do
  echo "== round 1 tree" >> $L
  (cd _r1 && SQ_EXT_FEATHER_ONLY=1 timeout -k 10 300 python tools/ext_probe.py 2>&1 | grep -E "feather ->" >> $L) || exit 1
  echo "== default, no plane groups" >> $L
  SQ_EXT_FLAGS=4 SQ_EXT_FEATHER_ONLY=1 timeout -k 10 300 python tools/ext_probe.py >> $L 2>&1 || exit 1
  for v in fb8 fb16 default; do
    echo "== $v" >> $L
    if [ $v = default ]; then SQ_EXT_FEATHER_ONLY=1 timeout -k 10 300 python tools/ext_probe.py >> $L 2>&1 || exit 1
    else SQ_EXT_FEATHER_ONLY=1 SQ_LIB_PATH=$PWD/$CS/libsquidstitch_$v.so timeout -k 10 300 python tools/ext_probe.py >> $L 2>&1 || exit 1; fi
  done
done
grep -E "^==|feather ->" $L
