#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/mid
mkdir -p $O
for k in "" $(ls scratch/lib_ko3*.so 2>/dev/null); do
  n=$(basename "${k:-shipped}" .so)
  export VH_LIB_PATH=${k:+$PWD/$k}
  [ -z "$k" ] && unset VH_LIB_PATH
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o t -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra-legs --preroll-seconds 0.05 --event-stride 1000 > $O/$n.json 2> $O/$n.err
  python3 - $O/$n/t_kernel_stats.csv $n <<'PY'
import csv,sys
out=[]
for r in csv.DictReader(open(sys.argv[1])):
    for k in ("k_render","k_compute_normals","k_integrate_fused"):
        if k in r["Name"] and int(r["Calls"])>100: out.append((k, round(float(r["AverageNs"])/1e3,1)))
print(sys.argv[2], out)
PY
done
