#!/bin/bash
# separate allocations vs one interleaved arena, one process each, on whatever box this call gets
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3/exp_interleaved_arena_box_$(date +%s).log
: > $O
for layout in separate interleaved separate interleaved; do
  echo "== layout $layout, headline batch" >> $O
  SQ_PROBE_LAYOUT=$layout timeout -k 10 400 python tools/order_probe.py 32 1 10 1 2>&1 | grep -E "round" >> $O
  echo "== layout $layout, config 3" >> $O
  SQ_PROBE_LAYOUT=$layout timeout -k 10 400 python tools/order_probe.py 16 4 10 1 2>&1 | grep -E "round" >> $O
done
cat $O
