#!/bin/bash
# r04 final: the evidence of the round's last build on ONE box: full GPU suite, smoke, the driver's bench command
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $O/gputest_final.log 2>&1 || { echo GPU suite failed; tail -30 $O/gputest_final.log; exit 1; }
tail -2 $O/gputest_final.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke_final.log 2>&1 || { echo smoke failed; tail -20 $O/smoke_final.log; exit 1; }
tail -1 $O/smoke_final.log
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_v45.out 2> $O/bench_v45.err || { echo bench failed; tail -30 $O/bench_v45.err; exit 1; }
grep "^{" $O/bench_v45.out | tail -1 > $O/bench_v45.json
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r4/bench_v45.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'launch', d['roofline']['launch_ms'], 'traffic', d['roofline']['traffic'])
print('arena', d['config']['memory']['canvas_arena']['class_slices'], d['config']['memory']['canvas_arena']['create_ms'])
print('parity', d['parity']['fused_mismatched_voxels'], d['parity']['shift_rmse_px'])
f = d['feather']
for k in ('u16', 'f32'):
    print('feather', k, f[k]['launch_ms'], f[k]['frac'], f[k]['parity']['mismatched_voxels'], f[k]['parity']['max_rel_err'])
h = d['headline_job_on_this_gpu']
print('job', h['value'], h['ms_per_step'], h['wall_ms_per_job'], h['roofline_frac'], h['host_ms_per_job'])
PY
