#!/bin/bash
# r04 q: inactive lanes clamped inside the row's own vectors (SQ_CLAMP_INSIDE) against the shipped clamp, one process each (alternating),
# canvas in the arena: 20 planes of config 3 through tools/arena_probe.py, then the bench's 40-plane launch
O=gpurun_out/r4; mkdir -p $O
for v in clamp shipped clamp shipped; do
  if [ $v = clamp ]; then export SQ_LIB_PATH=$PWD/image-stitcher_amd/csrc/libsquidstitch_clamp.so; else unset SQ_LIB_PATH; fi
  echo "== $v"; timeout -k 10 400 python3 tools/arena_probe.py 16 4 5 2 2>&1 | grep "arena:\|canvas mixed slots spread groups dealt" | cut -c1-240
done > $O/clamp_inside.log 2>&1
for v in clamp shipped; do
  if [ $v = clamp ]; then export SQ_LIB_PATH=$PWD/image-stitcher_amd/csrc/libsquidstitch_clamp.so; else unset SQ_LIB_PATH; fi
  SQ_BENCH_NO_REFERENCE_JOB=1 timeout -k 10 600 python3 bench.py --steps 10 --warmup 3 --no-live-traffic --no-feather --no-cpu-baseline 2>/dev/null | grep "^{" | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['roofline']['launch_ms'], d['roofline']['frac'], d['value'], d['config']['memory']['canvas_arena']['class_slices'])" >> $O/clamp_inside.log
done
cat $O/clamp_inside.log
