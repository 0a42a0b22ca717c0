#!/bin/bash
# r04 b: (1) how many placement classes the whole device memory falls into and how big they are (membw_gains 600);
# (2) the memory-side counters of the same-class pair / other-class pair / single plane row fill (membw_gains 500 under
# rocprofv3 --pmc, counters only, program directly after --; one pass per counter set)
O=gpurun_out/r4; mkdir -p $O
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 tools/membw_gains 3 0 0 1 600 256 > $O/classes_whole_memory.log 2>&1 || { echo classes failed; tail -5 $O/classes_whole_memory.log; exit 1; }
cat $O/classes_whole_memory.log
pass() {   # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $O/pmc_$name -o run -- tools/membw_gains 3 0 0 1 500 128 > $O/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $O/pmc_$name.log; return 1; }
  echo "== pass $name: $*" >> $O/placement_counters.log
  grep -E "^round|same-class plane" $O/pmc_$name.log >> $O/placement_counters.log
  python3 tools/r04/pmc_by_kernel.py $O/pmc_$name k_pair k_single >> $O/placement_counters.log
}
rm -f $O/placement_counters.log
pass wr1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum &&
pass wr2 TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_sum &&
pass tcc TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_SRC_FIFO_FULL_sum &&
pass lat TCC_WRITE_REQ_LATENCY_sum TCC_WRITE_REQ_sum TCC_LATENCY_FIFO_FULL_sum TCC_IB_STALL_sum &&
pass inst TCC_EA0_WRREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL
cat $O/placement_counters.log
find $O -name "*.csv" -size +20M -delete
