#!/bin/bash
# r04 m: the driver's command on the current build (canvas arena, feather leg, headline job, live traffic), with the arena trace
O=gpurun_out/r4; mkdir -p $O
SQ_ARENA_TRACE=1 timeout -k 10 1100 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_v35.json 2> $O/bench_v35.err || { echo bench failed; tail -30 $O/bench_v35.err; exit 1; }
grep "sq_arena\] [0-9]" $O/bench_v35.err | cut -c1-200; python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r4/bench_v35.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'launch', d['roofline']['launch_ms'], 'traffic', d['roofline']['traffic'])
print('arena', d['config']['memory']['canvas_arena'])
print('parity', d['parity'])
f = d['feather']
for k in ('u16', 'f32'):
    print('feather', k, f[k]['launch_ms'], f[k]['frac'], f[k].get('parity', {}).get('mismatched_voxels'), f[k].get('parity', {}).get('max_rel_err'))
h = d['headline_job_on_this_gpu']
print('job', h['value'], h['ms_per_step'], h['wall_ms_per_job'], h['roofline_frac'], h['host_ms_per_job'])
print('scale', d['scale_value'], d['scale_wall_ms_per_job'])
PY
