#!/bin/bash
# per-kernel times and SQ counters of the registration kernels on the headline job's batch
set -o pipefail
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/r3; mkdir -p $O
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/reg_trace -o run -- python3 $R/tools/reg_batch.py 3 > $O/reg_trace.log 2>&1 || { tail -5 $O/reg_trace.log; exit 1; }
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/reg_pmc1 -o run -- python3 $R/tools/reg_batch.py 1 > $O/reg_pmc1.log 2>&1 || { tail -5 $O/reg_pmc1.log; exit 1; }
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $O/reg_pmc2 -o run -- python3 $R/tools/reg_batch.py 1 > $O/reg_pmc2.log 2>&1 || { tail -5 $O/reg_pmc2.log; exit 1; }
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_F64 SQ_WAVES_EQ_64 -d $O/reg_pmc3 -o run -- python3 $R/tools/reg_batch.py 1 > $O/reg_pmc3.log 2>&1 || { tail -5 $O/reg_pmc3.log; }
rocprofv3 --pmc FETCH_SIZE -d $O/reg_pmc4 -o run -- python3 $R/tools/reg_batch.py 1 > $O/reg_pmc4.log 2>&1 || { tail -5 $O/reg_pmc4.log; }
rocprofv3 --pmc WRITE_SIZE -d $O/reg_pmc5 -o run -- python3 $R/tools/reg_batch.py 1 > $O/reg_pmc5.log 2>&1 || { tail -5 $O/reg_pmc5.log; }
cd $R
python3 - <<'PY'
import csv, glob, collections
O = 'gpurun_out/r3'
for f in sorted(glob.glob(O + '/reg_trace/*kernel_stats.csv')):
    for r in list(csv.DictReader(open(f)))[:12]:
        print(r['Name'][:70], r['Calls'], '%.3f ms avg' % (float(r['AverageNs']) / 1e6))
for d in ('reg_pmc1', 'reg_pmc2', 'reg_pmc3', 'reg_pmc4', 'reg_pmc5'):
    for f in sorted(glob.glob(f'{O}/{d}/*counter_collection.csv')):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:48] + '|' + r['Grid_Size'] + '|' + r['LDS_Block_Size']
            agg[k][r['Counter_Name']] += float(r['Counter_Value'])
            meta[k] = (r.get('VGPR_Count'), r.get('Workgroup_Size'))
        for k, v in agg.items():
            if 'rows_' in k or 'columns' in k or 'upsample_rows' in k:
                print(d, k, meta[k], {a: '%.4g' % b for a, b in v.items()})
PY
