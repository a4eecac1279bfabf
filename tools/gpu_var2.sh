#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/var2
mkdir -p $O
for cfg in cfg2 cfg3 cfg4; do
for k in "" $(ls scratch/lib_*.so 2>/dev/null); do
  n=$(basename "${k:-shipped}" .so)
  export VH_LIB_PATH=${k:+$PWD/$k}
  [ -z "$k" ] && unset VH_LIB_PATH
  timeout -k 10 300 python bench.py --config $cfg --no-streaming --steps 200 --warmup 40 --no-cpu-baseline --no-extra-legs > $O/$cfg.$n.json 2> $O/$cfg.$n.err
  python - "$O/$cfg.$n.json" "$cfg $n" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "render", j["roofline"]["avg_launch_us"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
done
