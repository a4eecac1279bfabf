#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r02ag}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_integrate_shapes.py -m gpu -x -q --timeout 120 > $O/pytest.log 2>&1; rc=$?
tail -15 $O/pytest.log
