#!/bin/bash
# round-2 run 5: full GPU suite, bench (cfg3), headline job rehearsals (1 rank, RCCL with one rank, 2 gloo ranks), rocprof
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r2_tests5.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/r2_tests5.log
timeout -k 10 600 python bench.py --steps 5 --warmup 2 > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err; echo "bench rc $?"; cat gpurun_out/r2_bench1.json; tail -3 gpurun_out/r2_bench1.err
SQ_BENCH_BREAKDOWN=1 timeout -k 10 600 python bench.py --workload cfg4 --planes 24 --steps 2 --warmup 1 > gpurun_out/r2_bench_cfg4_n1.json 2> gpurun_out/r2_bench_cfg4_n1.err; echo "cfg4 rc $?"; cat gpurun_out/r2_bench_cfg4_n1.json; tail -3 gpurun_out/r2_bench_cfg4_n1.err
SQ_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --workload cfg4 --planes 8 --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --sha-out gpurun_out/r2_sha_n1 > gpurun_out/r2_bench_cfg4_rccl1.json 2> gpurun_out/r2_bench_cfg4_rccl1.err; echo "rccl-1 rc $?"; cat gpurun_out/r2_bench_cfg4_rccl1.json; tail -3 gpurun_out/r2_bench_cfg4_rccl1.err
SQ_DIST_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --planes 8 --batch 2 --steps 1 --warmup 1 --sha-out gpurun_out/r2_sha_n2 > gpurun_out/r2_bench_cfg4_gloo2.json 2> gpurun_out/r2_bench_cfg4_gloo2.err; echo "gloo-2 rc $?"; cat gpurun_out/r2_bench_cfg4_gloo2.json; tail -3 gpurun_out/r2_bench_cfg4_gloo2.err
python - <<'PY'
import json
a = json.load(open('gpurun_out/r2_sha_n1.rank0'))
b = {}
for r in (0, 1):
    b.update(json.load(open(f'gpurun_out/r2_sha_n2.rank{r}')))
print('digests 1 rank vs 2 ranks:', 'IDENTICAL' if a == b and len(a) == 8 else f'DIFFER {a} {b}')
PY
