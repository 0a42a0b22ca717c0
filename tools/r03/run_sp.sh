#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3/exp_store_policy_r3.log
: > $O
for v in "" _sp1 _sp3 _sp4 ""; do
  echo "== libsquidstitch$v.so (SQ_STORE_POLICY: '' = nt (shipped), 1 = nt sc1, 3 = sc0 sc1, 4 = plain)" >> $O
  SQ_PROBE_LAYOUT=interleaved SQ_LIB_PATH=image-stitcher_amd/csrc/libsquidstitch$v.so timeout -k 10 300 python tools/order_probe.py 16 4 10 1 2>&1 | grep -E "spread groups dealt" >> $O
done
cat $O
