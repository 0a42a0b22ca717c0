#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp_address_axis.log
: > $O
for k in 1 2; do echo "== process $k" >> $O; timeout -k 10 300 tools/membw_gains 3 0 0 1 300 >> $O 2>&1 || exit 1; done
cat $O
