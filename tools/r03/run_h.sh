#!/bin/bash
# fast vs slow canvas placements under the TLB counters (program directly after --)
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3
rm -rf $O/pmc_tlb $O/pmc_ea
timeout -k 10 300 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum --kernel-trace --output-format csv -d $O/pmc_tlb -- tools/membw_gains 2 0 0 1 7 > $O/exp_placement_pmc_tlb.log 2>&1 || { echo pmc tlb failed; tail -5 $O/exp_placement_pmc_tlb.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum --kernel-trace --output-format csv -d $O/pmc_ea -- tools/membw_gains 2 0 0 1 7 > $O/exp_placement_pmc_ea.log 2>&1 || { echo pmc 2 failed; tail -5 $O/exp_placement_pmc_ea.log; exit 1; }
python3 - <<'PY'
import csv, glob, os
for d, log in (('gpurun_out/r3/pmc_tlb', 'gpurun_out/r3/exp_placement_pmc_tlb.log'), ('gpurun_out/r3/pmc_ea', 'gpurun_out/r3/exp_placement_pmc_ea.log')):
    print(open(log).read())
    f = glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    by = {}
    for r in rows:
        if 'k_regs' not in r['Kernel_Name']:
            continue
        by.setdefault(int(r['Dispatch_Id']), {'k': 'A' if 'Lb1' in r['Kernel_Name'] else 'P'})[r['Counter_Name']] = float(r['Counter_Value'])
    ids = sorted(by)
    for i, d_ in enumerate(ids):
        print(i, d_, by[d_])
    # keep the table small: drop the raw dirs' big files
PY
