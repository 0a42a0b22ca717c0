#!/bin/bash
# exp15: the headline job with contiguous plane runs per rank and batches of 10 (full groups of 5); RCCL branch at one rank;
# two gloo ranks sharing the card: per-plane digests equal to one rank's; distributed stitcher tests
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu > gpurun_out/r2_exp15_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r2_exp15_tests.log
SQ_BENCH_BREAKDOWN=1 timeout -k 10 500 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_exp15_job_n1.json 2> gpurun_out/r2_exp15_job_n1.err; echo "job rc $?"; cat gpurun_out/r2_exp15_job_n1.json; grep "host ms" gpurun_out/r2_exp15_job_n1.err
SQ_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --workload cfg4 --planes 10 --batch 5 --steps 1 --warmup 1 --no-cpu-baseline --sha-out gpurun_out/r2_sha_n1 > gpurun_out/r2_exp15_rccl1.json 2> gpurun_out/r2_exp15_rccl1.err; echo "rccl-1 rc $?"; tail -2 gpurun_out/r2_exp15_rccl1.err
SQ_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --planes 10 --batch 5 --steps 1 --warmup 1 --sha-out gpurun_out/r2_sha_n2 > gpurun_out/r2_exp15_gloo2.json 2> gpurun_out/r2_exp15_gloo2.err; echo "gloo-2 rc $?"; cat gpurun_out/r2_exp15_gloo2.json; tail -3 gpurun_out/r2_exp15_gloo2.err
python - <<'PY'
import json
a = json.load(open('gpurun_out/r2_sha_n1.rank0'))
b = {}
for r in (0, 1):
    b.update(json.load(open(f'gpurun_out/r2_sha_n2.rank{r}')))
print('planes', sorted(map(int, a)), 'two-rank digests equal one-rank digests:', a == b)
PY
