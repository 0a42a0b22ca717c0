#!/bin/bash
# r04 i: feather item order, experiment build, one process each (the canvas sits in a mixed arena):
#   0 span order | 3 canvas raster (bands of 16 rows, left to right) | 2 the overwrite plan's order: buckets of 8 TILE rows dealt to
#   the XCD lanes (every XCD works on a few gain rows at a time: the gain image is fetched into an L2 once, not once per tile)
O=gpurun_out/r4; mkdir -p $O
export SQ_LIB_PATH=$PWD/image-stitcher_amd/csrc/libsquidstitch_exp.so
for o in 0 3 2; do echo "== SQ_FEATHER_ORDER=$o"; SQ_FEATHER_ORDER=$o timeout -k 10 400 python3 tools/feather_probe.py 4 10 5 || exit 1; done > $O/feather_order.log 2>&1 || { echo failed; tail -20 $O/feather_order.log; exit 1; }
grep -v amdgpu.ids $O/feather_order.log
