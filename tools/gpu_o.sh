#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r02aa}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 120 > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc=$rc" | tee -a $O/summary.txt
tail -4 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs > $O/bench.json 2> $O/bench.err
timeout -k 10 300 python bench.py --config cfg3 --no-streaming --steps 200 --warmup 30 --no-cpu-baseline --no-extra-legs > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout -k 10 300 python bench.py --config cfg4 --steps 200 --warmup 30 --no-cpu-baseline --no-extra-legs > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout -k 10 300 python bench.py --config cfg3 --scene S2 --no-streaming --steps 100 --warmup 20 --no-cpu-baseline --no-extra-legs > $O/bench_dense.json 2> $O/bench_dense.err
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "render", j["roofline"]["avg_launch_us"], j["roofline"]["stage_us_warmup"])
    except Exception as e:
        print(f, "unreadable", e)
PY
