#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r02am}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 200 > $O/pytest_gpu.log 2>&1; rc=$?
tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/bench_integrate.py --gc 2>&1 | tail -1
VH_LIB_PATH=$PWD/scratch/lib_ko9.so timeout -k 10 200 python tools/bench_integrate.py --gc 2>&1 | tail -1
timeout -k 10 300 python bench.py --config cfg3 --scene S2 --no-streaming --steps 100 --warmup 20 --no-cpu-baseline --no-extra-legs > $O/dense.json 2> $O/dense.err
timeout -k 10 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs > $O/bench.json 2> $O/bench.err
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), j["roofline"]["stage_us_warmup"], j.get("rooflines",{}).get("integrate",{}).get("frac"))
    except Exception as e:
        print(f, "unreadable", e)
PY
