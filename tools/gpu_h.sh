#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02l
mkdir -p $O
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_full_20_5.json 2> $O/bench_full_20_5.err; echo "bench full rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs --event-stride 1 > $O/bench_stride1.json 2> $O/bench_stride1.err; echo "bench stride1 rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs --event-stride 1000 > $O/bench_stride1000.json 2> $O/bench_stride1000.err; echo "bench stride1000 rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02l/*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "host", j.get("host_enqueue_us_per_frame"), "render", j["roofline"]["avg_launch_us"], "pair", j["roofline"]["event_pair_us"], "oh", j["roofline"]["event_pair_overhead_us"], j["roofline"]["stage_us_warmup"])
        for k in ("value_with_upload","upload"):
            if k in j: print("   ", k, json.dumps(j[k])[:300])
    except Exception as e:
        print(f, "unreadable", e)
PY
