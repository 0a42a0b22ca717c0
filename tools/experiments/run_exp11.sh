#!/bin/bash
# round-2 experiment 11: allocation order (physical placement) -- tiles first vs canvas first, fresh process each
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r2_exp11.log
: > $L
for rep in 1 2 3 4; do
  echo "== tiles first" >> $L; timeout -k 10 300 python tools/fuse_probe.py --steps 6 --planes 20 --nflats 2 --flat f32 >> $L 2>&1 || exit 1
  echo "== canvas first" >> $L; timeout -k 10 300 python tools/fuse_probe.py --steps 6 --planes 20 --nflats 2 --flat f32 --canvas-first >> $L 2>&1 || exit 1
done
grep -E "^==|fuse:|tiles at" $L
