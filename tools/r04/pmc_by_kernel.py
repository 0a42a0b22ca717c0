#!/usr/bin/env python3
"""Mean Counter_Value per (kernel name, counter) of a rocprofv3 --pmc CSV directory; per-instance counters (dimension columns) are
summed per dispatch first, and their spread over the instances (min / max of the per-instance sums over the dispatches) is printed.
    python3 tools/r04/pmc_by_kernel.py DIR [name-substring ...]"""
import csv, glob, os, sys
from collections import defaultdict

def main():
    d = sys.argv[1]
    want = sys.argv[2:]
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        print('no counter_collection.csv under', d); return 1
    per_dispatch = defaultdict(float)       # (kernel, counter, dispatch) -> sum over instances
    per_instance = defaultdict(float)       # (kernel, counter, instance key) -> sum over dispatches
    ndisp = defaultdict(set)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                full = r['Kernel_Name']
                if want and not any(w in full for w in want):
                    continue
                k = full.replace('void ', '').replace('(anonymous namespace)::', '')
                k = k.split('(')[0].strip() or full
                c, v, disp = r['Counter_Name'], float(r['Counter_Value']), r.get('Dispatch_Id', r.get('Correlation_Id', '0'))
                per_dispatch[(k, c, disp)] += v
                ndisp[(k, c)].add(disp)
                inst = tuple((h, r[h]) for h in r if h.upper().startswith('DIMENSION') or h in ('Instance', 'XCC'))
                per_instance[(k, c, inst)] += v
    agg = defaultdict(list)
    for (k, c, disp), v in per_dispatch.items():
        agg[(k, c)].append(v)
    for (k, c) in sorted(agg):
        vals = agg[(k, c)]
        line = f'{k:52s} {c:36s} n={len(vals):3d} mean {sum(vals) / len(vals):16.1f}  min {min(vals):16.1f}  max {max(vals):16.1f}'
        insts = [v / len(ndisp[(k, c)]) for (kk, cc, i), v in per_instance.items() if kk == k and cc == c and i]
        if len(insts) > 1:
            line += f'  | {len(insts)} instances: min {min(insts):.1f} max {max(insts):.1f}'
        print(line)
    return 0

if __name__ == '__main__':
    sys.exit(main())
