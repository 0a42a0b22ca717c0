#!/bin/bash
# r04 u: where the feather kernel's time goes: experiment builds that run only the items 0 / 1 / 2 / 3-4 tiles cover (SQ_FEATHER_ONLY),
# one process each, config-3 geometry, canvas in the arena (tools/feather_probe.py 4 10 3)
O=gpurun_out/r4; mkdir -p $O
: > $O/feather_only.log
for k in 0 1 2 3; do
  echo "=== only items covered by $k tile(s) (3 = three or four)" >> $O/feather_only.log
  FEATHER_COVER=1 SQ_LIB_PATH=image-stitcher_amd/csrc/libsquidstitch_only$k.so timeout -k 10 250 python3 tools/feather_probe.py 4 10 3 2>&1 | grep -v amdgpu.ids | sed -n '3,8p' >> $O/feather_only.log || { echo probe $k failed; tail -5 $O/feather_only.log; exit 1; }
done
cat $O/feather_only.log
