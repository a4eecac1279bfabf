#!/bin/bash
# the shipped library and every scratch/lib_*.so variant through the bench on one box (twice, interleaved)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-var3}
CFGS=${2:-cfg2}
mkdir -p $O
for rep in 1 2; do
for cfg in $CFGS; do
for k in "" $(ls scratch/lib_*.so 2>/dev/null); do
  n=$(basename "${k:-shipped}" .so)
  export VH_LIB_PATH=${k:+$PWD/$k}
  [ -z "$k" ] && unset VH_LIB_PATH
  timeout -k 10 300 python bench.py --config $cfg --no-streaming --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs > $O/$cfg.$n.$rep.json 2> $O/$cfg.$n.$rep.err
  python - "$O/$cfg.$n.$rep.json" "$cfg $n" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "render", j["roofline"]["avg_launch_us"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
done
done
