#!/bin/bash
# exp23: seam owners in the per-plane pipeline (no gains / gains one plane at a time): parity suite, then against the
# previous build ("old") in one process on the same buffers
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_fuse_gpu.py tests/test_c_abi_gpu.py -x -q -m gpu > gpurun_out/r2_exp23_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r2_exp23_tests.log; [ $rc = 0 ] || exit 1
{
echo "== no gains, 16 planes: check + old / default"
timeout -k 10 300 python tools/fuse_probe.py --grid 4 --planes 3 --steps 2 --check
timeout -k 10 300 python tools/fuse_probe.py --planes 16 --steps 5 --libs old,default
echo "== float32 gains, one plane at a time (flags 4), 8 planes: old / default (6 waves asked) / 5 waves"
timeout -k 10 300 python tools/fuse_probe.py --grid 4 --planes 3 --flat f32 --flags 4 --steps 2 --check
timeout -k 10 300 python tools/fuse_probe.py --planes 8 --nflats 2 --flat f32 --flags 4 --steps 5 --libs old,default,w5
echo "== float64 gains, one plane at a time (flags 4), 8 planes"
timeout -k 10 300 python tools/fuse_probe.py --planes 8 --nflats 2 --flat f64 --flags 4 --steps 4 --libs old,default
echo "== the grouped kernel (unchanged code path): old / default"
timeout -k 10 300 python tools/fuse_probe.py --planes 20 --nflats 2 --flat f32 --steps 4 --libs old,default
} > gpurun_out/r2_exp23.log 2>&1
grep -v amdgpu.ids gpurun_out/r2_exp23.log
