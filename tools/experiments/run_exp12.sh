#!/bin/bash
# exp12: decomposition of the row-segment copy (which side, which alignment) + the job's fixed host overhead
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 300 ./tools/membw 8 2d > gpurun_out/r2_exp12_membw2d.log 2>&1

cat gpurun_out/r2_exp12_membw2d.log

