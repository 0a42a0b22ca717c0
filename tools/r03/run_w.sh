#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
SQ_PROBE_UNIT_MINOR=1 SQ_LIB_PATH=image-stitcher_amd/csrc/libsquidstitch_uminor.so timeout -k 10 500 python tools/order_probe.py 16 4 10 2 > $O/exp_unit_minor.log 2>&1; echo "rc $?"; cat $O/exp_unit_minor.log
SQ_PROBE_UNIT_MINOR=1 SQ_LIB_PATH=image-stitcher_amd/csrc/libsquidstitch_uminor.so timeout -k 10 500 python tools/order_probe.py 32 1 10 2 > $O/exp_unit_minor_cfg4.log 2>&1; echo "rc $?"; cat $O/exp_unit_minor_cfg4.log
