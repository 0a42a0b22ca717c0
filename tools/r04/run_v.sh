#!/bin/bash
# r04 v: the driver's bench command alone (the arenas closed before the counter passes; the passes' own errors in the log)
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_v38.out 2> $O/bench_v38.err || { echo bench failed; tail -30 $O/bench_v38.err; exit 1; }
grep "^{" $O/bench_v38.out | tail -1 > $O/bench_v38.json
grep "live traffic" $O/bench_v38.err | cut -c1-1800
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r4/bench_v38.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'launch', d['roofline']['launch_ms'], 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'][:40])
print('arena', d['config']['memory']['canvas_arena']['class_slices'], d['config']['memory']['canvas_arena']['create_ms'], d['config']['memory']['canvas_arena']['n_candidates'])
f = d['feather']
for k in ('u16', 'f32'):
    print('feather', k, f[k]['launch_ms'], f[k]['frac'], f[k]['parity']['mismatched_voxels'], f[k]['parity']['max_rel_err'])
h = d['headline_job_on_this_gpu']
print('job', h['value'], h['ms_per_step'], h['wall_ms_per_job'], h['roofline_frac'], h['host_ms_per_job'])
PY
