#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_arena.json 2> $O/bench_arena.err; echo "bench rc $?"; cat $O/bench_arena.json; tail -3 $O/bench_arena.err
timeout -k 10 600 python bench.py --gpus 1 --steps 10 --warmup 3 --layout separate --no-cpu-baseline > $O/bench_separate.json 2> $O/bench_separate.err; echo "bench separate rc $?"; cut -c1-300 $O/bench_separate.json; python3 -c "
import json; d=json.load(open('$O/bench_separate.json')); print('separate: frac', d['roofline']['frac'], 'job', d['headline_job_on_this_gpu']['value'], d['headline_job_on_this_gpu']['roofline_frac'])
d=json.load(open('$O/bench_arena.json')); print('arena: frac', d['roofline']['frac'], 'job', d['headline_job_on_this_gpu']['value'], d['headline_job_on_this_gpu']['roofline_frac'])"
SQ_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --planes 8 --batch 2 --steps 1 --warmup 1 --sha-out $O/sha_arena_n2 > $O/bench_arena_n2.json 2> $O/bench_arena_n2.err; echo "n2 rc $?"
timeout -k 10 300 python bench.py --workload cfg4 --planes 8 --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --layout separate --sha-out $O/sha_sep_n1 > /dev/null 2>&1; echo "n1 rc $?"
python3 - <<'PY'
import json
a = json.load(open('gpurun_out/r3/sha_sep_n1.rank0'))
b = {}
for r in range(2):
    b.update(json.load(open(f'gpurun_out/r3/sha_arena_n2.rank{r}')))
assert a == b and len(a) == 8, (a, b)
print('digests of 8 planes: two self-launched gloo ranks with the arena layout == one rank with separate allocations')
PY
