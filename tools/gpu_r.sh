#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r02ah}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 200 > $O/pytest_gpu.log 2>&1; rc=$?
tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
bash tools/gpu_var2.sh ${1:-r02ah}_var "cfg2 cfg3 cfg4"
