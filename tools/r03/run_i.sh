#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3
: > $O/exp_placement2.log
for k in 1 2; do echo "== process $k" >> $O/exp_placement2.log; timeout -k 10 300 tools/membw_gains 3 0 0 1 7 >> $O/exp_placement2.log 2>&1 || exit 1; done
cat $O/exp_placement2.log
for t in 1 2 4 8 16; do echo "-- SQ_PLAN_THREADS=$t"; SQ_LIB_PATH=image-stitcher_amd/csrc/libsquidstitch_planexp.so SQ_PLAN_TIMING=1 SQ_PLAN_THREADS=$t python tools/plan_time.py 2>&1 | tail -4; done > $O/exp_plan_threads.log 2>&1
cat $O/exp_plan_threads.log
