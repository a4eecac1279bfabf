#!/bin/bash
# first measurement of round 2: new tests, then the bench the way the driver runs it and its variants
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02a
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_frame_loop.py -x -q > $O/pytest_loop.log 2>&1; echo "pytest loop rc=$?" | tee -a $O/summary.txt
for v in "" "--no-alloc-ahead" "--python-loop" "--preroll-seconds 0"; do
  n=$(echo "$v" | tr -d ' -')
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs $v > $O/bench_20_5_$n.json 2> $O/bench_20_5_$n.err; echo "bench 20/5 [$v] rc=$?" | tee -a $O/summary.txt
done
for v in "" "--no-alloc-ahead" "--python-loop"; do
  n=$(echo "$v" | tr -d ' -')
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs --stages $v > $O/bench_900_$n.json 2> $O/bench_900_$n.err; echo "bench 900/100 [$v] rc=$?" | tee -a $O/summary.txt
done
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_full_20_5.json 2> $O/bench_full_20_5.err; echo "bench full rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02a/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "host", j.get("host_enqueue_us_per_frame"), "render", j["roofline"]["avg_launch_us"], j["roofline"]["stage_us_warmup"])
    except Exception as e:
        print(f, "unreadable", e)
PY
