#!/bin/bash
# r04 y: (1) what the separate pass over the row ends of the grouped blends costs (experiment build without it, SQ_FEATHER_NO_EDGES)
# against the shipped library, one process each; (2) the registration probe with a long-line case
O=gpurun_out/r4; mkdir -p $O
: > $O/feather_noedges.log
for lib in libsquidstitch_noedges.so libsquidstitch.so; do
  echo "=== $lib" >> $O/feather_noedges.log
  SQ_LIB_PATH=image-stitcher_amd/csrc/$lib timeout -k 10 250 python3 tools/feather_probe.py 4 10 3 2>&1 | grep -v amdgpu.ids | sed -n '3,6p' >> $O/feather_noedges.log || { echo probe failed; tail -5 $O/feather_noedges.log; exit 1; }
done
cat $O/feather_noedges.log
timeout -k 10 600 python3 tools/kernel_probe.py registration > $O/kernel_probe_y.log 2>&1 || { echo probe failed; tail -20 $O/kernel_probe_y.log; exit 1; }
grep -i "pairs/s" $O/kernel_probe_y.log
