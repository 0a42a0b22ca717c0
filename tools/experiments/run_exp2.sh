#!/bin/bash
# round-2 experiment 2: does the ORDER in which memory is walked explain one-shot 0.79 vs persistent 0.61?
set -o pipefail
mkdir -p gpurun_out
CS=image-stitcher_amd/csrc
timeout -k 10 500 tools/membw 8 > gpurun_out/r2_membw2.log 2>&1 || { echo membw failed; tail -5 gpurun_out/r2_membw2.log; exit 1; }
grep -E "atomic|streams|rows" gpurun_out/r2_membw2.log
: > gpurun_out/r2_exp2.log
for rep in 1 2; do
for order in 2 0 3; do
  for flat in none f32; do
    echo "== order $order flat $flat" >> gpurun_out/r2_exp2.log
    SQ_PLAN_ORDER=$order SQ_LIB_PATH=$PWD/$CS/libsquidstitch_exp.so timeout -k 10 300 python tools/fuse_probe.py --planes 16 --nflats 4 --flat $flat >> gpurun_out/r2_exp2.log 2>&1 || exit 1
  done
done
done
grep -E "^==|fuse:" gpurun_out/r2_exp2.log
