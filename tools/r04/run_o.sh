#!/bin/bash
# r04 o: the chunk files through the native writer: the store tests, then the end-to-end split probe again
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_blosc_gpu.py tests/test_configs_gpu.py tests/test_stitcher_gpu.py -x -q > $O/test_store.log 2>&1 || { echo tests failed; tail -40 $O/test_store.log; exit 1; }
tail -2 $O/test_store.log
timeout -k 10 1000 python3 tools/e2e_split_probe.py /tmp 4 > $O/e2e_split2.log 2>&1 || { echo failed; tail -30 $O/e2e_split2.log; exit 1; }
grep -v amdgpu.ids $O/e2e_split2.log
