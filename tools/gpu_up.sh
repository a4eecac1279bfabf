#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/up
mkdir -p $O
timeout -k 10 200 python -m pytest tests/test_gpu_frame_loop.py tests/test_c_abi.py -x -q --timeout 60 > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for k in "" $(ls scratch/lib_up*.so 2>/dev/null); do
  n=$(basename "${k:-shipped}" .so)
  export VH_LIB_PATH=${k:+$PWD/$k}
  [ -z "$k" ] && unset VH_LIB_PATH
  timeout -k 10 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs --frames-on-host > $O/$n.json 2> $O/$n.err
  python - "$O/$n.json" "$n" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "render", j["roofline"]["avg_launch_us"], "wait", j["host_wait_us_per_frame"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
