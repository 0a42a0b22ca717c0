#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3/exp_interleaved_arena.log
: > $O
for rep in 1 2; do for layout in separate interleaved; do
  echo "== process: layout $layout, config 3 (40 planes)" >> $O
  SQ_PROBE_LAYOUT=$layout timeout -k 10 400 python tools/order_probe.py 16 4 10 1 2>&1 | grep -E "planes \(|round" >> $O
  echo "== process: layout $layout, headline batch (10 planes of the 32x32 grid)" >> $O
  SQ_PROBE_LAYOUT=$layout timeout -k 10 400 python tools/order_probe.py 32 1 10 1 2>&1 | grep -E "planes \(|round" >> $O
done; done
cat $O
