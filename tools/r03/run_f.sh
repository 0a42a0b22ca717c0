#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp_placement.log
: > $O
for k in 1 2 3; do
  echo "== process $k" >> $O
  timeout -k 10 200 tools/membw_gains 3 0 0 1 9 >> $O 2>&1 || { echo failed; tail -3 $O; exit 1; }
done
cat $O
