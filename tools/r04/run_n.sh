#!/bin/bash
# r04 n: end to end from files, split into its stages, on the box's disk and on /dev/shm (tools/e2e_split_probe.py)
O=gpurun_out/r4; mkdir -p $O
df -h /tmp /dev/shm 2>&1 | tee $O/e2e_split.log
timeout -k 10 1000 python3 tools/e2e_split_probe.py /tmp 4 >> $O/e2e_split.log 2>&1 || { echo failed; tail -30 $O/e2e_split.log; exit 1; }
grep -v amdgpu.ids $O/e2e_split.log
