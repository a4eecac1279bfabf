#!/bin/bash
# end of round: the whole GPU suite, smoke(), then the evidence run (tools/collect_profile.sh)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 200 > gpurun_out/final_pytest_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/final_pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python __graft_entry__.py smoke > gpurun_out/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/final_smoke.log
bash tools/collect_profile.sh > gpurun_out/final_stdout.log 2>&1
bash tools/bench_lines.sh final_lines
cp gpurun_out/final_lines/bench.json gpurun_out/final/bench.json
cp gpurun_out/final_lines/bench_driver_20_5.json gpurun_out/final/bench_driver_20_5.json
tail -3 gpurun_out/final_stdout.log
