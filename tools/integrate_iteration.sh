#!/bin/bash
# integrate-kernel iteration: the tests that cover it, then the kernel alone on the dense scene and the dense bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-integrate_iteration}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_integrate_shapes.py tests/test_gpu_parity.py tests/test_gpu_frame_loop.py -m gpu -x -q --timeout 120 > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 200 python tools/bench_integrate.py > $O/integ.txt 2>&1; tail -2 $O/integ.txt
timeout -k 10 200 python tools/bench_integrate.py --gc > $O/integ_gc.txt 2>&1; tail -2 $O/integ_gc.txt
timeout -k 10 300 python bench.py --config cfg3 --scene S2 --no-streaming --steps 100 --warmup 20 --no-cpu-baseline --no-extra-legs > $O/dense.json 2> $O/dense.err
timeout -k 10 300 python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs > $O/bench.json 2> $O/bench.err
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/*.json")):
    try:
        j=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), j["roofline"]["stage_us_warmup"], j.get("rooflines",{}).get("integrate"))
    except Exception as e:
        print(f, "unreadable", e)
PY
