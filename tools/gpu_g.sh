#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02j
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_integrate_shapes.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_frame_loop.py -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc=$rc" | tee -a $O/summary.txt
tail -5 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/bench_integrate.py 2>&1 | tail -1 | tee -a $O/ko.txt
timeout -k 10 200 python tools/bench_integrate.py --gc 2>&1 | tail -1 | tee -a $O/ko.txt
for k in $(ls scratch/lib_ko*.so); do
  VH_LIB_PATH=$PWD/$k timeout -k 10 200 python tools/bench_integrate.py 2>&1 | tail -1 | tee -a $O/ko.txt
done
