#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r3/counters_avail.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -i -E "utcl|tlb|translation|xnack" gpurun_out/r3/counters_avail.txt | cut -c1-200 | sort -u | head -60
