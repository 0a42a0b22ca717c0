#!/bin/bash
# r04 k2: the counters of the feather kernels on the last build (after the corners went through the plane groups): FETCH_SIZE, WRITE_SIZE
# and the SQ view, one pass each, program after --; then the end-to-end split probe on the last build
O=gpurun_out/r4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
pass() { local name=$1; shift
  rm -rf $O/feather2_pmc_$name
  timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $O/feather2_pmc_$name -o run -- python3 tools/feather_probe.py 4 10 2 > $O/feather2_pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $O/feather2_pmc_$name.log; return 1; }
  echo "== pass $name: $*" >> $O/feather_counters2.log
  python3 tools/r04/pmc_by_kernel.py $O/feather2_pmc_$name fuse_feather >> $O/feather_counters2.log
  rm -rf $O/feather2_pmc_$name; }
rm -f $O/feather_counters2.log
pass fetch FETCH_SIZE && pass write WRITE_SIZE && pass sq1 SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES || exit 1
cat $O/feather_counters2.log
timeout -k 10 600 python3 tools/e2e_split_probe.py /tmp 4 > $O/e2e_split3.log 2>&1 || { echo e2e failed; tail -30 $O/e2e_split3.log; exit 1; }
grep -v amdgpu.ids $O/e2e_split3.log | tail -25
