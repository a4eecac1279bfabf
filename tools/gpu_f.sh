#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02g
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_integrate_shapes.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc=$rc" | tee -a $O/summary.txt
tail -5 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense -o t -- python3 bench.py --config cfg3 --scene S2 --no-streaming --steps 40 --warmup 10 --no-cpu-baseline --no-extra-legs --preroll-seconds 0.05 > $O/dense.json 2> $O/dense.err; echo "dense rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/r02g/dense/t_kernel_stats.csv")):
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
