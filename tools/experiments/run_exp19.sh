#!/bin/bash
# exp19: tile files read straight into the page-locked staging buffer (tiffio.read_image_into): end-to-end probe + the from-files tests
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_stitcher_gpu.py tests/test_configs_gpu.py -x -q -m gpu > gpurun_out/r2_exp19_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r2_exp19_tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 600 python tools/e2e_probe.py > gpurun_out/r2_exp19_e2e.log 2>&1; echo "e2e rc $?"; grep -v amdgpu.ids gpurun_out/r2_exp19_e2e.log
