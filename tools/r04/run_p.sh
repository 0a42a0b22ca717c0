#!/bin/bash
# r04 p: the N-rank paths with the canvas arena: two self-launched gloo ranks on the one GPU (digests against one rank), the RCCL
# branch with one rank, and smoke()
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo smoke failed; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 500 python3 bench.py --workload cfg4 --planes 8 --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --sha-out $O/sha_one > $O/bench_job_one_rank.json 2> $O/bench_job_one_rank.err || { echo one-rank job failed; tail -20 $O/bench_job_one_rank.err; exit 1; }
SQ_DIST_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --planes 8 --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --sha-out $O/sha_two > $O/bench_job_gloo_two_ranks.json 2> $O/bench_job_gloo_two_ranks.err || { echo two-rank job failed; tail -30 $O/bench_job_gloo_two_ranks.err; exit 1; }
python3 - <<'PY'
import json
one = json.load(open('gpurun_out/r4/sha_one.rank0'))
two = {}
for r in (0, 1):
    two.update(json.load(open(f'gpurun_out/r4/sha_two.rank{r}')))
assert one == two and len(one) == 8, (one, two)
d = [json.loads(l) for l in open('gpurun_out/r4/bench_job_gloo_two_ranks.json') if l.startswith('{')][-1]
print('two gloo ranks on one GPU: digests of all 8 planes equal one rank\'s; parallelism:', d['config']['parallelism'][:120], '| scale_value', d['scale_value'], '| wall', d['wall_ms_per_job'])
PY
SQ_BENCH_FORCE_DIST=1 timeout -k 10 500 python3 bench.py --workload cfg4 --planes 10 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_job_rccl_one_rank.json 2> $O/bench_job_rccl_one_rank.err || { echo rccl one-rank failed; tail -20 $O/bench_job_rccl_one_rank.err; exit 1; }
python3 -c "
import json; d=[json.loads(l) for l in open('gpurun_out/r4/bench_job_rccl_one_rank.json') if l.startswith('{')][-1]; print('RCCL, one rank:', d['value'], d['roofline']['frac'], d['config']['memory']['canvas_arena']['class_slices'])"
