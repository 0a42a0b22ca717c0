#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
rm -rf $O/prof_reg
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/prof_reg -- python3 tools/kernel_probe.py registration > $O/kernel_probe_under_rocprof.log 2>&1 || { tail -5 $O/kernel_probe_under_rocprof.log; exit 1; }
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3/prof_reg/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# group consecutive dispatches into registration calls by init_tables_kernel
calls, cur = [], None
for r in rows:
    n = r['Kernel_Name']
    if 'init_tables_kernel' in n:
        cur = {}
        calls.append(cur)
    if cur is not None and 'anonymous' in n:
        import re
        short = re.search(r'(\w+_kernel)', n).group(1)
        cur[short] = cur.get(short, 0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for i, c in enumerate(calls):
    print(i, ' '.join(f'{k}={v:.0f}us' for k, v in c.items()))
PY
rm -rf $O/prof_reg
