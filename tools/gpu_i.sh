#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02m
mkdir -p $O
timeout -k 10 600 python bench.py --config cfg3 --steps 200 --warmup 30 --no-cpu-baseline --no-extra-legs --stages > $O/cfg3_stream.json 2> $O/cfg3_stream.err; echo "cfg3 streaming rc=$?" | tee -a $O/summary.txt
timeout -k 10 600 python bench.py --config cfg3 --steps 200 --warmup 30 --no-cpu-baseline --no-extra-legs --no-streaming --stages > $O/cfg3_nostream.json 2> $O/cfg3_nostream.err; echo "cfg3 no streaming rc=$?" | tee -a $O/summary.txt
timeout -k 10 600 python bench.py --config cfg4 --steps 200 --warmup 30 --no-cpu-baseline --no-extra-legs --stages > $O/cfg4.json 2> $O/cfg4.err; echo "cfg4 rc=$?" | tee -a $O/summary.txt
tail -3 $O/*.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02m/*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "host", j.get("host_enqueue_us_per_frame"), "wait", j.get("host_wait_us_per_frame"), "render", j["roofline"]["avg_launch_us"], j["roofline"]["stage_us_warmup"], j["roofline"]["blocks_in_frustum"], j.get("streaming"))
    except Exception as e:
        print(f, "unreadable", e)
PY
