#!/bin/bash
# r04 g: the bench's feather leg (uint16 + gains, float32 out) on the mixed arena, short run
O=gpurun_out/r4; mkdir -p $O
SQ_BENCH_NO_REFERENCE_JOB=1 timeout -k 10 900 python3 bench.py --steps 5 --warmup 2 --no-live-traffic "$@" > $O/bench_feather.json 2> $O/bench_feather.err || { echo bench failed; tail -30 $O/bench_feather.err; exit 1; }
tail -3 $O/bench_feather.err; python3 -c "
import json; d=json.load(open('$O/bench_feather.json')); print(json.dumps(d['feather'], indent=1)); print(d['value'], d['roofline']['frac'])"
