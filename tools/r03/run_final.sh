#!/bin/bash
# final evidence of the round on ONE box: full GPU suite, smoke, the default bench line, the 2-rank self-launched rehearsal
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests_final.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -6 $O/tests_final.log; [ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc $?"; cat $O/bench_final.json
SQ_DIST_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --planes 20 --batch 5 --steps 2 --warmup 1 > $O/bench_final_n2_gloo_one_gpu.json 2> $O/bench_final_n2.err; echo "n2 rc $?"; cut -c1-600 $O/bench_final_n2_gloo_one_gpu.json
