#!/bin/bash
# LDS / wait counters of the registration kernels on a 992-pair batch of 1024 x 256 crops (the job's batch)
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3
cat > /tmp/reg_batch.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from image_stitcher_amd import native, registration
dev = torch.device('cuda:0')
g, T = 32, 2048
tiles = torch.randint(0, 65535, (g * g // 4, T, T), dtype=torch.int32, device=dev).to(torch.uint16)   # 256 tiles, pairs index them cyclically
mm = native.tile_minmax(tiles)
(hp, (h0, h1)), (vp, (v0, v1)) = registration.all_pairs(g, g, T, T, 256, 256)
for batch in (hp, vp):
    batch['ref_tile'] %= len(tiles); batch['mov_tile'] %= len(tiles)
for rep in range(2):
    for pairs, n0, n1 in ((hp, h0, h1), (vp, v0, v1)):
        native.register_pairs_async(tiles, mm, pairs, n0, n1, 10, native.SQ_NORM_PHASE).fetch()
print('done', len(hp), len(vp))
PY
rm -rf $O/pmc_reg1 $O/pmc_reg2
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/pmc_reg1 -- python3 /tmp/reg_batch.py > $O/pmc_reg1.log 2>&1 || { tail -5 $O/pmc_reg1.log; exit 1; }
python3 - <<'PY'
import csv, glob, re
f = glob.glob('gpurun_out/r3/pmc_reg1/**/*_counter_collection.csv', recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    m = re.search(r'(\w+_kernel)', r['Kernel_Name'])
    if not m or 'anonymous' not in r['Kernel_Name']:
        continue
    d = acc.setdefault(m.group(1), {})
    d.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    d['t'] = d.get('t', []) + [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3]
    d['vgpr'] = r['VGPR_Count']; d['lds'] = r['LDS_Block_Size']
for k, d in acc.items():
    if k in ('rows_forward_kernel', 'columns_kernel', 'rows_inverse_kernel', 'upsample_rows_kernel', 'minmax_kernel'):
        mean = lambda n: sum(d[n]) / len(d[n]) if n in d else float('nan')
        print(f"{k}: {mean('t'):.0f} us (under the counters)  vgpr {d['vgpr']} lds {d['lds']}  LDS insts {mean('SQ_INSTS_LDS'):.3g}  bank-conflict cycles / active LDS cycles "
              f"{mean('SQ_LDS_BANK_CONFLICT') / max(mean('SQ_ACTIVE_INST_LDS'), 1):.2f}  wait-LDS / wave cycles {mean('SQ_WAIT_INST_LDS') / mean('SQ_WAVE_CYCLES'):.2f}  "
              f"wait-any / wave cycles {mean('SQ_WAIT_ANY') / mean('SQ_WAVE_CYCLES'):.2f}  VALU insts {mean('SQ_INSTS_VALU'):.3g}")
PY
rm -rf $O/pmc_reg1
