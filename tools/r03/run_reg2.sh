#!/bin/bash
# registration after a kernel change: its tests, the 30-pair probe, per-kernel times + HBM fetch of the headline batch
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_register_gpu.py tests/test_stitcher_gpu.py tests/test_distributed_gpu.py -m gpu -x -q > $O/reg2_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -4 $O/reg2_tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 300 python tools/kernel_probe.py registration > $O/reg2_probe.log 2>&1; echo "probe rc $?"; grep "pairs/s" $O/reg2_probe.log | cut -c1-160
cd /tmp
rm -rf $O/reg_trace $O/reg_pmc4
rocprofv3 --kernel-trace --stats -d $O/reg_trace -o run -- python3 $R/tools/reg_batch.py 3 > $O/reg_trace.log 2>&1 || { tail -5 $O/reg_trace.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE -d $O/reg_pmc4 -o run -- python3 $R/tools/reg_batch.py 1 > $O/reg_pmc4.log 2>&1 || { tail -5 $O/reg_pmc4.log; }
cd $R
python3 - <<'PY'
import sqlite3, collections
c = sqlite3.connect('gpurun_out/r3/reg_trace/run_results.db')
d = collections.defaultdict(list)
for r in c.execute("select name, duration, grid_x, grid_y, workgroup_x, lds_size from kernels order by start"):
    d[(r[0][20:70],) + tuple(r[2:])].append(r[1] / 1e6)
tot = 0
for k, v in d.items():
    if 'at::native' in k[0]: continue
    print(k, len(v), ' '.join('%.2f' % x for x in v[-3:]))
c = sqlite3.connect('gpurun_out/r3/reg_pmc4/run_results.db')
agg = collections.defaultdict(float)
for r in c.execute("select kernel_name, grid_size_x, counter_name, value from counters_collection"):
    agg[(r[0][26:60], r[1], r[2])] += r[3]
for k, v in agg.items():
    if any(s in k[0] for s in ('rows_', 'columns', 'upsample_rows')): print(k, '%.4g KB x2 = %.2f GB' % (v, v * 2 / 1e6))
PY
