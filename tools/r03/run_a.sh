#!/bin/bash
# round 3, call A: new distributed tests, N-rank bench through its own launcher (gloo, one GPU), default bench
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
timeout -k 10 500 python -m pytest tests/test_distributed_gpu.py tests/test_stitcher_gpu.py -x -q -m gpu > $O/a_tests.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -5 $O/a_tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 200 python tests/golden/make_blosc_golden.py dump $O/blosc_device_frames.npz || exit 1
# one rank, then two gloo ranks started by bench.py itself: digests must agree
timeout -k 10 300 python bench.py --workload cfg4 --planes 8 --batch 2 --steps 2 --warmup 1 --no-cpu-baseline --sha-out $O/sha_n1 > $O/a_job_n1.json 2> $O/a_job_n1.err; echo "job n1 rc $?"; cat $O/a_job_n1.json
SQ_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --planes 8 --batch 2 --steps 2 --warmup 1 --sha-out $O/sha_n2 > $O/a_job_n2.json 2> $O/a_job_n2.err; rc=$?; echo "job n2 (self-launched) rc $rc"; cat $O/a_job_n2.json; [ $rc = 0 ] || { tail -20 $O/a_job_n2.err; exit 1; }
python - <<'PY'
import json
a = json.load(open('gpurun_out/r3/sha_n1.rank0'))
b = {}
for r in range(2):
    b.update(json.load(open(f'gpurun_out/r3/sha_n2.rank{r}')))
assert a == b and len(a) == 8, (a, b)
print('digests of 8 planes: two self-launched gloo ranks == one rank')
PY
[ $? = 0 ] || exit 1
SQ_BENCH_BREAKDOWN=1 timeout -k 10 600 python bench.py --steps 10 --warmup 3 > $O/a_bench.json 2> $O/a_bench.err; echo "bench rc $?"; cat $O/a_bench.json; tail -3 $O/a_bench.err
# the two structural experiments on the fusion kernel's inner work, membw first
timeout -k 10 300 tools/membw_gains 5 > $O/exp_lds_membw.log 2>&1; echo "membw_gains rc $?"; cat $O/exp_lds_membw.log
