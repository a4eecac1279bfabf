#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02d
mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest gpu rc=$rc" | tee -a $O/summary.txt
tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_full_20_5.json 2> $O/bench_full_20_5.err; echo "bench full rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra-legs --stages > $O/bench_900.json 2> $O/bench_900.err; echo "bench 900 rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense -o t -- python3 bench.py --config cfg3 --scene S2 --no-streaming --steps 40 --warmup 10 --no-cpu-baseline --no-extra-legs --preroll-seconds 0.05 > $O/dense.json 2> $O/dense.err; echo "dense rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json,glob,csv
for f in sorted(glob.glob("gpurun_out/r02d/*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "host", j.get("host_enqueue_us_per_frame"), "render", j["roofline"]["avg_launch_us"], j["roofline"]["stage_us_warmup"], j["roofline"]["blocks_in_frustum"])
        for k in ("rooflines","value_with_upload","upload"):
            if k in j: print("   ", k, json.dumps(j[k])[:900])
    except Exception as e:
        print(f, "unreadable", e)
for r in csv.DictReader(open("gpurun_out/r02d/dense/t_kernel_stats.csv")):
    print(r["Name"][:60], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
