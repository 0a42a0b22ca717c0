#!/bin/bash
# round-2 experiment 1: box yardstick (membw), GPU tests after the ADVICE fixes, reciprocal-table variants
set -o pipefail
mkdir -p gpurun_out
CS=image-stitcher_amd/csrc
timeout -k 10 400 tools/membw 8 > gpurun_out/r2_membw.log 2>&1 || { echo membw failed; tail -5 gpurun_out/r2_membw.log; exit 1; }
echo "membw done"
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests1.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2_tests1.log
: > gpurun_out/r2_exp1.log
SQ_LIB_PATH=$PWD/$CS/libsquidstitch_v1w4d2.so timeout -k 10 300 python tools/fuse_probe.py --grid 4 --planes 2 --flat f32 --rcp --check >> gpurun_out/r2_exp1.log 2>&1 || exit 1
for rep in 1 2; do
  echo "== base" >> gpurun_out/r2_exp1.log
  timeout -k 10 300 python tools/fuse_probe.py --planes 16 --nflats 4 --flat f32 >> gpurun_out/r2_exp1.log 2>&1 || exit 1
  for v in v1w4d2 v1w5d1 v1w4d3; do
    echo "== $v" >> gpurun_out/r2_exp1.log
    SQ_LIB_PATH=$PWD/$CS/libsquidstitch_$v.so timeout -k 10 300 python tools/fuse_probe.py --planes 16 --nflats 4 --flat f32 --rcp >> gpurun_out/r2_exp1.log 2>&1 || exit 1
  done
  echo "== base noflat" >> gpurun_out/r2_exp1.log
  timeout -k 10 300 python tools/fuse_probe.py --planes 16 --flat none >> gpurun_out/r2_exp1.log 2>&1 || exit 1
done
grep -E "^==|fuse:|mismatched" gpurun_out/r2_exp1.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2_bench0.json 2> gpurun_out/r2_bench0.err; echo "bench rc $?"; cat gpurun_out/r2_bench0.json
