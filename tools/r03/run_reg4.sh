#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_register_gpu.py tests/test_stitcher_gpu.py tests/test_distributed_gpu.py -m gpu -x -q > $O/reg4_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -2 $O/reg4_tests.log; [ $rc = 0 ] || exit 1
L=image-stitcher_amd/csrc/libsquidstitch_regexp.so
{
python tools/reg_time.py
SQ_LIB_PATH=$L SQ_REG_TC=4 SQ_REG_COL_THREADS=512 python tools/reg_time.py
python tools/reg_time.py
} 2>&1 | grep -v amdgpu.ids > $O/exp_registration_shapes2.log
cat $O/exp_registration_shapes2.log | cut -c1-200
timeout -k 10 300 python tools/kernel_probe.py registration > $O/reg4_probe.log 2>&1; echo "probe rc $?"; grep "pairs/s" $O/reg4_probe.log | cut -c1-160
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
    if any(s in k[0] for s in ('rows_', 'columns', 'upsample_rows', 'minmax_k')): print(k, len(v), ' '.join('%.2f' % x for x in v[-3:]))
PY
