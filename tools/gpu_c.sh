#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02q
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc=$rc" | tee -a $O/summary.txt
tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
for v in "a:" "b:--no-alloc-ahead"; do
  n=${v%%:*}; f=${v#*:}
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs $f > $O/bench_20_5_$n.json 2> $O/bench_20_5_$n.err; echo "bench 20/5 [$f] rc=$?" | tee -a $O/summary.txt
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs --stages $f > $O/bench_900_$n.json 2> $O/bench_900_$n.err; echo "bench 900/100 [$f] rc=$?" | tee -a $O/summary.txt
done
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_full_20_5.json 2> $O/bench_full_20_5.err; echo "bench full rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extra-legs --preroll-seconds 0.05 > $O/trace.json 2> $O/trace.err; echo "trace rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02q/bench_*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "host", j.get("host_enqueue_us_per_frame"), "render", j["roofline"]["avg_launch_us"], j["roofline"]["stage_us_warmup"])
        for k in ("rooflines","value_with_upload","upload"):
            if k in j: print("   ", k, json.dumps(j[k])[:600])
    except Exception as e:
        print(f, "unreadable", e)
PY
