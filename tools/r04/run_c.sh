#!/bin/bash
# r04 c: a MIXED arena (HIP virtual memory management: physical slices classified by the pair fill, mapped round-robin over the
# classes) against a plain hipMalloc of the same size, same process (tools/membw_gains mode 700), at several slice sizes
O=gpurun_out/r4; mkdir -p $O
for slice in "$@"; do
  timeout -k 10 400 tools/membw_gains 3 0 0 1 700 96 $slice > $O/mixed_arena_$slice.log 2>&1 || { echo failed $slice; tail -20 $O/mixed_arena_$slice.log; exit 1; }
  echo "=== slice $slice MiB"; cat $O/mixed_arena_$slice.log
done
