#!/bin/bash
# r04 j: feather after the row-wise float32 one-tile path and with raster order by default: tests, then the probe (shipped lib)
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_fuse_gpu.py tests/test_stitcher_gpu.py tests/test_plan_gpu.py -x -q -k "feather or groups or queues or blend or plan" > $O/test_feather2.log 2>&1 || { echo tests failed; tail -60 $O/test_feather2.log; exit 1; }
tail -3 $O/test_feather2.log
timeout -k 10 400 python3 tools/feather_probe.py 4 10 5 > $O/feather_probe_40.log 2>&1 || { echo failed; tail -20 $O/feather_probe_40.log; exit 1; }
grep -v amdgpu.ids $O/feather_probe_40.log
