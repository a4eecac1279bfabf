#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-bench_lines}
mkdir -p $O
timeout -k 10 500 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_20_5.json 2> $O/bench_driver_20_5.err; echo "bench 20/5 rc=$?"
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench*.json")):
    j=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], j["value"], j.get("value_with_upload"), j.get("upload",{}).get("upload_us"), j["roofline"].get("second_bound"))
PY
