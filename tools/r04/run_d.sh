#!/bin/bash
# r04 d: the real fusion kernel with its canvas in a mixed arena vs a plain allocation, one process (tools/arena_probe.py)
O=gpurun_out/r4; mkdir -p $O
SQ_ARENA_TRACE=1 timeout -k 10 600 python3 tools/arena_probe.py 16 4 5 3 > $O/arena_probe_cfg3.log 2>&1 || { echo failed; tail -30 $O/arena_probe_cfg3.log; exit 1; }
cat $O/arena_probe_cfg3.log
