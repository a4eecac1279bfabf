#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r02y}
mkdir -p $O
for k in $(ls scratch/lib_*.so); do
  VH_LIB_PATH=$PWD/$k timeout -k 10 200 python tools/bench_integrate.py --gc 2>&1 | tail -1 | tee -a $O/ko.txt
done
