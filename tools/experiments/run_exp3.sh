#!/bin/bash
# round-2 experiment 3: why is a one-shot copy faster than a persistent one (store acknowledgements? order?);
# plane groups (gains + reciprocals once per group of z planes)
set -o pipefail
mkdir -p gpurun_out
CS=image-stitcher_amd/csrc
timeout -k 10 600 tools/membw 8 > gpurun_out/r2_membw3.log 2>&1 || { echo membw failed; tail -5 gpurun_out/r2_membw3.log; exit 1; }
grep -E "vmcnt|WORKGROUP|loader|one-shot, blocks dealt over 1 " gpurun_out/r2_membw3.log
timeout -k 10 600 python -m pytest tests/test_fuse_gpu.py -x -q > gpurun_out/r2_tests3.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2_tests3.log
: > gpurun_out/r2_exp3.log
timeout -k 10 300 python tools/fuse_probe.py --grid 4 --planes 7 --flat f32 --check >> gpurun_out/r2_exp3.log 2>&1 || exit 1
for rep in 1 2; do
  echo "== per-plane kernel (flags 4)" >> gpurun_out/r2_exp3.log
  timeout -k 10 300 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 --flags 4 >> gpurun_out/r2_exp3.log 2>&1 || exit 1
  echo "== zg ZB=5" >> gpurun_out/r2_exp3.log
  timeout -k 10 300 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 >> gpurun_out/r2_exp3.log 2>&1 || exit 1
  for v in zb2 zb3 zb5w6 zb7; do
    echo "== zg $v" >> gpurun_out/r2_exp3.log
    SQ_LIB_PATH=$PWD/$CS/libsquidstitch_$v.so timeout -k 10 300 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 >> gpurun_out/r2_exp3.log 2>&1 || exit 1
  done
done
grep -E "^==|fuse:|mismatched" gpurun_out/r2_exp3.log
