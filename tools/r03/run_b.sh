#!/bin/bash
# plane-stride sweep of the membw_gains structures: do the 5 planes' streams collide in the DRAM mapping?
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3/exp_plane_stride.log
: > $O
for pads in "0 0" "256 0" "0 256" "4352 4352" "65792 65792" "1048832 1048832" "16777472 16777472" "2048 2048" "8192 8192" "0 1664" "33554688 0"; do
  echo "== src_pad dst_pad = $pads" >> $O
  timeout -k 10 120 tools/membw_gains 5 $pads 1 >> $O 2>&1 || { echo failed; tail -3 $O; exit 1; }
done
grep -E "^==|^A |^P " $O
