#!/bin/bash
# r04 t: the refined-reciprocal quotients with ONE correction (tools/div_probe.hip): the fusion tests (exhaustive divide selftests,
# feather parity), then the feather probe on the shipped library
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_fuse_gpu.py -x -q > $O/test_fuse_t.log 2>&1 || { echo tests failed; tail -60 $O/test_fuse_t.log; exit 1; }
tail -3 $O/test_fuse_t.log
timeout -k 10 400 python3 tools/feather_probe.py 4 10 > $O/feather_probe_t.log 2>&1 || { echo probe failed; tail -30 $O/feather_probe_t.log; exit 1; }
cat $O/feather_probe_t.log
