#!/bin/bash
# r04 z: is the one-off failure of test_plane_groups_on_line_aligned_canvases[True-0-1] (5 voxels of one plane at 2 d + 1) the edge
# restructure's (experiment build "edges") or older?  The fusion test file, whole, three times per library.
O=gpurun_out/r4; mkdir -p $O
: > $O/flake_hunt.log
for lib in libsquidstitch_edges.so libsquidstitch.so; do
  for k in 1 2 3; do
    echo "=== $lib run $k" >> $O/flake_hunt.log
    SQ_LIB_PATH=image-stitcher_amd/csrc/$lib timeout -k 10 400 python3 -m pytest tests/test_fuse_gpu.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED|Mismatched|Max abs|Max rel|plane [0-9]+ \(" >> $O/flake_hunt.log
    echo "done $lib $k"
  done
done
cat $O/flake_hunt.log
