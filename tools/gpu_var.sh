#!/bin/bash
# the default bench under each measurement build of scratch/ (and the shipped one)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/var
mkdir -p $O
for k in "" $(ls scratch/lib_*.so 2>/dev/null); do
  n=$(basename "${k:-shipped}" .so)
  VH_LIB_PATH=${k:+$PWD/$k} timeout -k 10 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs ${VAR_ARGS} > $O/$n.json 2> $O/$n.err
  python - "$O/$n.json" "$n" <<'PY'
import json,sys
try:
    j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[2], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "render", j["roofline"]["avg_launch_us"], j["roofline"]["stage_us_warmup"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
done
