#!/bin/bash
# r04 h: feather after the packed / grouped-float32 rework: the fusion tests, then the bench's feather leg
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_fuse_gpu.py tests/test_stitcher_gpu.py -x -q -k "feather or groups or queues or blend" > $O/test_feather.log 2>&1 || { echo tests failed; tail -60 $O/test_feather.log; exit 1; }
tail -3 $O/test_feather.log
bash tools/r04/run_g.sh
