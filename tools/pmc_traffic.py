"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into profiles/pmc_traffic_latest.json.

    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write cfg3 40 "<note>" [out.json]
"""
import csv, glob, json, sys

fetch_dir, write_dir, workload, planes = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
note = sys.argv[5] if len(sys.argv) > 5 else ''
def mean(d, key):
    f = glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True)[0]
    v = [float(r['Counter_Value']) for r in csv.DictReader(open(f))
         if 'fuse_overwrite' in r['Kernel_Name'] and r['Counter_Name'] == key]
    return sum(v) / len(v), len(v)
fk, nf = mean(fetch_dir, 'FETCH_SIZE')
wk, nw = mean(write_dir, 'WRITE_SIZE')
out = dict(workload=workload, planes=planes, FETCH_SIZE_KB=fk, WRITE_SIZE_KB=wk, launches=[nf, nw],
           traffic_bytes_per_launch=(2 * fk + wk) * 1024,
           formula='(2*FETCH_SIZE + WRITE_SIZE)*1024: gfx950 reports half the bytes of wide coalesced reads '
                   '(MI355X_MICROARCH.md, HBM); Infinity-Cache hits are counted', note=note)
json.dump(out, open(sys.argv[6] if len(sys.argv) > 6 else 'profiles/pmc_traffic_latest.json', 'w'), indent=1)
print(json.dumps(out))
