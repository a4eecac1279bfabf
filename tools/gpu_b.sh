#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02b
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_frame_loop.py -x -q > $O/pytest_loop.log 2>&1; echo "pytest loop rc=$?" | tee -a $O/summary.txt
timeout -k 10 120 python tools/h2d_probe.py > $O/h2d_probe.log 2>&1; echo "h2d probe rc=$?" | tee -a $O/summary.txt
for v in "ahead:" "inorder:--no-alloc-ahead"; do
  n=${v%%:*}; f=${v#*:}
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$n -o t -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra-legs --preroll-seconds 0.05 $f > $O/trace_$n.json 2> $O/trace_$n.err; echo "trace $n rc=$?" | tee -a $O/summary.txt
done
ls -la $O/trace_ahead | head
