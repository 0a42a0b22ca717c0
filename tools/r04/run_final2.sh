#!/bin/bash
# r04 final2: after regions pipelined through a stream of their own in run(), and "the released memory is back before sq_arena_create returns": arena + C-caller tests, the driver's bench command
# (with the arena trace), then the rocprofv3 kernel stats of the bench and of the feather probe on the same box
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_arena_gpu.py tests/test_c_abi_gpu.py tests/test_stitcher_gpu.py tests/test_configs_gpu.py tests/test_distributed_gpu.py tests/test_pyramid_gpu.py -x -q > $O/arena_tests3.log 2>&1 || { echo arena tests failed; tail -30 $O/arena_tests3.log; exit 1; }
tail -1 $O/arena_tests3.log
SQ_ARENA_TRACE=1 timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_v40.out 2> $O/bench_v40.err || { echo bench failed; tail -30 $O/bench_v40.err; exit 1; }
grep "^{" $O/bench_v40.out | tail -1 > $O/bench_v40.json
grep "live traffic\|sq_arena\] \(release\|the released\|[0-9]* candidate\)" $O/bench_v40.err | cut -c1-300
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r4/bench_v40.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'frac', d['roofline']['frac'], 'launch', d['roofline']['launch_ms'], 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'][:40])
print('arena', d['config']['memory']['canvas_arena']['class_slices'], d['config']['memory']['canvas_arena']['create_ms'], d['config']['memory']['canvas_arena']['n_candidates'])
print('parity', d['parity']['fused_mismatched_voxels'], d['parity']['shift_rmse_px'])
f = d['feather']
for k in ('u16', 'f32'):
    print('feather', k, f[k]['launch_ms'], f[k]['frac'], f[k]['parity']['mismatched_voxels'], f[k]['parity']['max_rel_err'])
h = d['headline_job_on_this_gpu']
print('job', h['value'], h['ms_per_step'], h['wall_ms_per_job'], h['roofline_frac'], h['host_ms_per_job'])
PY
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export SQ_BENCH_NO_REFERENCE_JOB=1
rm -rf $O/bench_trace2 $O/feather_trace2
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_trace2 -o run -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-live-traffic --no-feather > $O/bench_under_rocprof2.json 2> $O/bench_under_rocprof2.err || { echo rocprof bench failed; tail -5 $O/bench_under_rocprof2.err; exit 1; }
cp $O/bench_trace2/run_kernel_stats.csv $O/bench_v40_kernel_stats.csv
grep "^{" $O/bench_under_rocprof2.json | tail -1 > $O/bench_v40_under_rocprof.json
head -4 $O/bench_v40_kernel_stats.csv | cut -c1-200
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/feather_trace2 -o run -- python3 tools/feather_probe.py 4 10 5 > $O/feather_under_rocprof2.log 2>&1 || { echo rocprof feather failed; tail -5 $O/feather_under_rocprof2.log; exit 1; }
cp $O/feather_trace2/run_kernel_stats.csv $O/feather_v40_kernel_stats.csv
head -6 $O/feather_v40_kernel_stats.csv | cut -c1-200
grep "planes" $O/feather_under_rocprof2.log | tail -4
rm -rf $O/bench_trace2 $O/feather_trace2
timeout -k 10 300 python3 tools/cfg5_probe.py 8 2 > $O/cfg5_probe_after2.log 2>&1 || { echo cfg5 probe failed; tail -5 $O/cfg5_probe_after2.log; exit 1; }
tail -2 $O/cfg5_probe_after2.log
