#!/bin/bash
mkdir -p gpurun_out/r3
for k in 1 2; do echo "== process $k"; timeout -k 10 200 tools/membw_gains 3 0 0 1 400 2>&1 | grep round; done > gpurun_out/r3/exp_zero_tail.log
cat gpurun_out/r3/exp_zero_tail.log
