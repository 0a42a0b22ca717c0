#!/bin/bash
# r04 x: registration lines too long for the LDS (transformed in scratch lines of the workspace): the registration tests, then the
# registration probe of round 3 on the unchanged LDS paths (the kernels were split into bodies: their rates must not have moved)
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_register_gpu.py -x -q > $O/test_register_x.log 2>&1 || { echo tests failed; tail -40 $O/test_register_x.log; exit 1; }
tail -3 $O/test_register_x.log
timeout -k 10 600 python3 tools/kernel_probe.py registration > $O/kernel_probe_x.log 2>&1 || { echo probe failed; tail -20 $O/kernel_probe_x.log; exit 1; }
grep -i "pairs/s\|regist" $O/kernel_probe_x.log | head -30
