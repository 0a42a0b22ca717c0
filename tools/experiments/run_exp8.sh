#!/bin/bash
# round-2 experiment 8: feather blend path -- taller blend items + loads of the next pair ahead of the blend
set -o pipefail
mkdir -p gpurun_out
CS=image-stitcher_amd/csrc
L=gpurun_out/r2_exp8.log
: > $L
timeout -k 10 600 python -m pytest tests/test_fuse_gpu.py -x -q -k "feather or fuzz" >> $L 2>&1; echo "pytest rc $?"; tail -3 $L
for rep in 1 2; do
  for v in fb8 fb16 default fb64; do
    echo "== $v" >> $L
    if [ $v = default ]; then SQ_EXT_FEATHER_ONLY=1 timeout -k 10 300 python tools/ext_probe.py >> $L 2>&1 || exit 1
    else SQ_EXT_FEATHER_ONLY=1 SQ_LIB_PATH=$PWD/$CS/libsquidstitch_$v.so timeout -k 10 300 python tools/ext_probe.py >> $L 2>&1 || exit 1; fi
  done
done
grep -E "^==|feather ->" $L
