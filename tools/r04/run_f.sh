#!/bin/bash
# r04 f: the arena tests, then the driver's bench command with the canvas in the mixed arena (the default layout now)
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_arena_gpu.py -x -q > $O/test_arena.log 2>&1 || { echo tests failed; tail -40 $O/test_arena.log; exit 1; }
tail -3 $O/test_arena.log
SQ_ARENA_TRACE=1 timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_mixed.json 2> $O/bench_mixed.err || { echo bench failed; tail -30 $O/bench_mixed.err; exit 1; }
tail -5 $O/bench_mixed.err; cat $O/bench_mixed.json
