#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_register_gpu.py tests/test_stitcher_gpu.py tests/test_distributed_gpu.py -m gpu -x -q > $O/reg3_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 $O/reg3_tests.log; [ $rc = 0 ] || exit 1
L=image-stitcher_amd/csrc/libsquidstitch_regexp.so
{
python tools/reg_time.py
SQ_LIB_PATH=$L python tools/reg_time.py
for tc in 4 2 1; do for th in 512 256 128; do SQ_LIB_PATH=$L SQ_REG_TC=$tc SQ_REG_COL_THREADS=$th timeout -k 10 120 python tools/reg_time.py || exit 1; done; done
for rl in 2 4 8 16; do SQ_LIB_PATH=$L SQ_REG_RLF=$rl timeout -k 10 120 python tools/reg_time.py || exit 1; done
for rl in 2 4 8 16; do SQ_LIB_PATH=$L SQ_REG_RLI=$rl timeout -k 10 120 python tools/reg_time.py || exit 1; done
} > $O/exp_registration_shapes.log 2>&1
cat $O/exp_registration_shapes.log | cut -c1-200
cd /tmp
rm -rf $O/reg_trace
rocprofv3 --kernel-trace --stats -d $O/reg_trace -o run -- python3 $R/tools/reg_batch.py 3 > $O/reg_trace.log 2>&1 || { tail -5 $O/reg_trace.log; exit 1; }
cd $R
python3 - <<'PY'
import sqlite3, collections
c = sqlite3.connect('gpurun_out/r3/reg_trace/run_results.db')
d = collections.defaultdict(list)
for r in c.execute("select name, duration, grid_x, grid_y, workgroup_x, lds_size from kernels order by start"):
    d[(r[0][20:70],) + tuple(r[2:])].append(r[1] / 1e6)
for k, v in d.items():
    if any(s in k[0] for s in ('rows_', 'columns', 'upsample_rows')): print(k, len(v), ' '.join('%.2f' % x for x in v[-3:]))
PY
