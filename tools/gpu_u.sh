#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r02ao}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_streaming.py tests/test_golden.py tests/test_gpu_hashgrid.py tests/test_gpu_fullsize.py tests/test_reconstruction.py -m gpu -x -q --timeout 200 > $O/pytest.log 2>&1; rc=$?
tail -12 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python bench.py --config cfg3 --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs > $O/cfg3_stream.json 2> $O/cfg3_stream.err
timeout -k 10 300 python bench.py --config cfg3 --steps 300 --warmup 50 --streaming-radius 1.2 --streaming-pos-z 1.6 --streaming-extent 0.5 --no-cpu-baseline --no-extra-legs > $O/cfg3_traffic.json 2> $O/cfg3_traffic.err
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "host wait", j.get("host_wait_us_per_frame"), j.get("streaming"))
    except Exception as e:
        print(f, "unreadable", e)
PY
