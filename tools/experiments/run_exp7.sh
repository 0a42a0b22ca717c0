#!/bin/bash
# round-2 experiment 7: the plane-group kernel as a ONE-SHOT launch (one workgroup per (group, item)) vs persistent
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r2_exp7.log
: > $L
run() { echo "== $*" >> $L; timeout -k 10 300 python tools/fuse_probe.py --steps 8 "$@" >> $L 2>&1 || exit 1; }
run --grid 4 --planes 7 --flat f32 --check --flags 2 --blocks 1000000000
for rep in 1 2; do
  run --planes 20 --nflats 2 --flat f32
  run --planes 20 --nflats 2 --flat f32 --flags 2
  run --planes 20 --nflats 2 --flat f32 --flags 2 --blocks 1000000000
  run --planes 20 --nflats 2 --flat f32 --flags 2 --blocks 4096
  run --planes 16 --flat none
  run --planes 16 --flat none --flags 2 --blocks 1000000000
done
grep -E "^==|fuse:|mismatched" $L
