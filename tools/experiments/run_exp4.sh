#!/bin/bash
# round-2 experiment 4: canvas-band lanes (order 4) vs tile-row-block lanes (order 2); 4-row items; plane-count sweep
set -o pipefail
mkdir -p gpurun_out
CS=image-stitcher_amd/csrc
L=gpurun_out/r2_exp4.log
: > $L
run() { # lib order args...
  lib=$1; order=$2; shift 2
  echo "== $lib order $order $*" >> $L
  SQ_PLAN_ORDER=$order SQ_LIB_PATH=$PWD/$CS/libsquidstitch_$lib.so timeout -k 10 300 python tools/fuse_probe.py --steps 8 "$@" >> $L 2>&1 || exit 1
}
run exp 4 --grid 4 --planes 7 --flat f32 --check
run exp 4 --grid 5 --planes 3 --flat none --check
for rep in 1 2; do
  run exp 2 --planes 16 --flat none
  run exp 4 --planes 16 --flat none
  run expr4 4 --planes 16 --flat none
  run exp 2 --planes 20 --nflats 2 --flat f32
  run exp 4 --planes 20 --nflats 2 --flat f32
  run expw6 2 --planes 20 --nflats 2 --flat f32
  run expw6 4 --planes 20 --nflats 2 --flat f32
  run expr4 4 --planes 20 --nflats 2 --flat f32
done
run exp 2 --planes 2 --flat none
run exp 2 --planes 32 --flat none
run exp 4 --planes 2 --flat none
run exp 4 --planes 32 --flat none
grep -E "^==|fuse:|mismatched" $L
