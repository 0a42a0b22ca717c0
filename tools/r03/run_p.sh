#!/bin/bash
# full GPU suite + smoke on HEAD, then the plane-group kernel at 4 (shipped) / 5 / 6 waves per SIMD under the new placement
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests_full.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -8 $O/tests_full.log; [ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
: > $O/exp_occupancy_r3.log
for v in "" _zg5 _zg6 ""; do
  echo "== libsquidstitch$v.so" >> $O/exp_occupancy_r3.log
  SQ_LIB_PATH=image-stitcher_amd/csrc/libsquidstitch$v.so timeout -k 10 300 python tools/order_probe.py 16 4 10 1 2>&1 | grep -E "spread groups dealt|plane  groups consecutive" >> $O/exp_occupancy_r3.log
done
cat $O/exp_occupancy_r3.log
