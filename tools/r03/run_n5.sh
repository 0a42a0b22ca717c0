#!/bin/bash
# five self-launched gloo ranks on the one GPU against one rank: the digests of all 24 fused planes must agree
set -o pipefail
mkdir -p gpurun_out/r3
O=gpurun_out/r3
rm -f $O/sha_n5.rank* $O/sha_n1b.rank*
timeout -k 10 400 python bench.py --workload cfg4 --planes 24 --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --sha-out $O/sha_n1b > $O/n1b.json 2> $O/n1b.err; echo "n1 rc $?"
SQ_DIST_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 5 --planes 24 --batch 2 --steps 1 --warmup 1 --sha-out $O/sha_n5 > $O/n5.json 2> $O/n5.err; echo "n5 rc $?"; tail -2 $O/n5.err | cut -c1-300
python - <<'PY'
import json, glob
one = json.load(open('gpurun_out/r3/sha_n1b.rank0'))
six = {}
for f in sorted(glob.glob('gpurun_out/r3/sha_n5.rank*')):
    six.update(json.load(open(f)))
print(len(one), 'planes on one rank,', len(six), 'over', len(glob.glob('gpurun_out/r3/sha_n5.rank*')), 'ranks; equal digests:', one == six)
d = json.loads(open('gpurun_out/r3/n5.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('n_gpus', 'value', 'ms_per_step', 'first_job_ms')}, d['config']['registration'][:120], d['parity'])
PY
