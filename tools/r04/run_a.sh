#!/bin/bash
# r04 a: what the box is (memory / compute partition mode), which counters rocprofv3 offers, and the class scan of
# tools/membw_gains mode 500 without a profiler (does one 128 GiB allocation still fall into classes?)
O=gpurun_out/r4; mkdir -p $O
{
  echo "== partition modes"
  for f in /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition \
           /sys/class/drm/card*/device/available_memory_partition /sys/class/drm/card*/device/mem_info_vram_total; do
    [ -r $f ] && echo "$f: $(cat $f)"
  done
  rocm-smi --showmemorypartition --showcomputepartition 2>&1 | head -20
  amd-smi static --partition 2>&1 | head -30
  rocminfo 2>&1 | grep -i -E "pool|size|segment|granule|xcc|compute unit|marketing" | head -40
} > $O/box.txt 2>&1
rocprofv3 -L > $O/counters_avail.txt 2>&1 || rocprofv3 --list-avail > $O/counters_avail.txt 2>&1
grep -o "TCC_[A-Z0-9_]*" $O/counters_avail.txt | sort -u > $O/tcc_counters.txt
wc -l $O/tcc_counters.txt
timeout -k 10 400 tools/membw_gains 3 0 0 1 500 128 > $O/classes_plain.log 2>&1 || { echo membw failed; tail -5 $O/classes_plain.log; exit 1; }
tail -8 $O/classes_plain.log
cat $O/box.txt | head -40
