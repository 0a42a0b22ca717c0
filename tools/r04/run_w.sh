#!/bin/bash
# r04 w: waves per SIMD asked for the grouped feather kernel (2 / 3 shipped / 4), one process each (tools/feather_probe.py 4 10 3)
O=gpurun_out/r4; mkdir -p $O
: > $O/feather_waves.log
for w in 2 4; do
  echo "=== SQ_WAVES_FEATHER_ZG=$w" >> $O/feather_waves.log
  SQ_LIB_PATH=image-stitcher_amd/csrc/libsquidstitch_fw$w.so timeout -k 10 250 python3 tools/feather_probe.py 4 10 3 2>&1 | grep -v amdgpu.ids | sed -n '3,6p' >> $O/feather_waves.log || { echo probe $w failed; tail -5 $O/feather_waves.log; exit 1; }
done
cat $O/feather_waves.log
