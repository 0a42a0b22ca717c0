#!/bin/bash
# exp25: the headline job with FOUR gloo ranks sharing the one card (16 planes, batches of 2): per-plane digests against one rank's
set -o pipefail
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -f gpurun_out/r2_sha4_* 
timeout -k 10 400 python bench.py --workload cfg4 --planes 16 --batch 2 --steps 1 --warmup 1 --no-cpu-baseline --sha-out gpurun_out/r2_sha4_n1 > gpurun_out/r2_exp25_n1.json 2> gpurun_out/r2_exp25_n1.err; echo "n1 rc $?"
SQ_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 4 --planes 16 --batch 2 --steps 1 --warmup 1 --sha-out gpurun_out/r2_sha4_n4 > gpurun_out/r2_exp25_gloo4.json 2> gpurun_out/r2_exp25_gloo4.err; echo "gloo-4 rc $?"; cat gpurun_out/r2_exp25_gloo4.json | head -c 1500; echo; tail -3 gpurun_out/r2_exp25_gloo4.err
python - <<'PY'
import json
a = json.load(open('gpurun_out/r2_sha4_n1.rank0'))
b = {}
sizes = []
for r in range(4):
    d = json.load(open(f'gpurun_out/r2_sha4_n4.rank{r}'))
    sizes.append(sorted(map(int, d)))
    b.update(d)
print('planes per rank', sizes)
print('four-rank digests equal one-rank digests:', a == b, len(a), len(b))
PY
