#!/bin/bash
mkdir -p gpurun_out/r3
for t in 1 8; do echo "-- SQ_PLAN_THREADS=$t"; SQ_LIB_PATH=image-stitcher_amd/csrc/libsquidstitch_planexp.so SQ_PLAN_TIMING=1 SQ_PLAN_THREADS=$t python tools/plan_time.py 2>&1 | tail -4; done > gpurun_out/r3/exp_plan_phases.log 2>&1
cat gpurun_out/r3/exp_plan_phases.log
