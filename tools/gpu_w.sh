#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r02ar}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 200 > $O/pytest_gpu.log 2>&1; rc=$?
tail -3 $O/pytest_gpu.log
[ $rc -ne 0 ] && exit 1
for c in cfg2 cfg4; do VH_LIB_PATH=$PWD/scratch/lib_ko42.so timeout -k 10 300 python tools/riders_stamps.py $c 2>/dev/null | tail -1 > $O/riders_$c.json; done
mv scratch/lib_ko42.so scratch/x_ko42.keep
for c in cfg2 cfg3 cfg4; do
  timeout -k 10 300 python bench.py --config $c --no-streaming --steps 300 --warmup 50 --no-cpu-baseline --no-extra-legs > $O/bench_$c.json 2> $O/bench_$c.err
done
python - $O <<'PY'
import json,glob,sys
for f in sorted(glob.glob(sys.argv[1]+"/bench*.json")):
    j=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], j["value"], "us/frame", round(1e3*j["ms_per_step"],1), "render", j["roofline"]["avg_launch_us"])
for c in ("cfg2","cfg4"):
    j=json.loads(open(sys.argv[1]+f"/riders_{c}.json").read())
    print(c, {k:(v["end_us"][-1] if isinstance(v,dict) else v) for k,v in j.items()})
PY
