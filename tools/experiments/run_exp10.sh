#!/bin/bash
# round-2 experiment 10: does the plane-group kernel gain from more resident waves?  (scalar-base addressing; full groups only
# builds at 5 / 6 / 7 waves per SIMD: 84 / 77 / 72 VGPRs)
set -o pipefail
mkdir -p gpurun_out
CS=image-stitcher_amd/csrc
L=gpurun_out/r2_exp10.log
: > $L
run() { lib=$1; shift; echo "== $lib $*" >> $L
  if [ $lib = default ]; then timeout -k 10 300 python tools/fuse_probe.py --steps 8 "$@" >> $L 2>&1 || exit 1
  else SQ_LIB_PATH=$PWD/$CS/libsquidstitch_$lib.so timeout -k 10 300 python tools/fuse_probe.py --steps 8 "$@" >> $L 2>&1 || exit 1; fi; }
run zf7 --grid 4 --planes 10 --nflats 2 --flat f32 --check
for rep in 1 2 3; do
  for v in default w6 zf1 zf6 zf7; do run $v --planes 20 --nflats 2 --flat f32; done
done
grep -E "^==|fuse:|mismatched" $L
