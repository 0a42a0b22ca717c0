#!/bin/bash
# full GPU suite + smoke on HEAD
set -o pipefail
mkdir -p gpurun_out/r3
export TMPDIR=/tmp
O=gpurun_out/r3
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/tests_full.log 2>&1; rc=$?; echo "pytest rc $rc"; tail -8 $O/tests_full.log; [ $rc = 0 ] || exit 1
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
