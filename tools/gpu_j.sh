#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02n
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_streaming.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -15 $O/pytest.log
timeout -k 10 600 python bench.py --config cfg3 --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs --streaming-radius 1.2 --streaming-pos-z 1.6 --streaming-extent 0.5 > $O/cfg3_stream_traffic.json 2> $O/cfg3_stream_traffic.err; echo "cfg3 streaming traffic rc=$?" | tee -a $O/summary.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02n/*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "host", j.get("host_enqueue_us_per_frame"), "wait", j.get("host_wait_us_per_frame"), j.get("streaming"))
    except Exception as e:
        print(f, "unreadable", e)
PY
