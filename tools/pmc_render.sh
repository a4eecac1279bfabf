#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_render
mkdir -p $O
for k in "" $(ls scratch/lib_*.so 2>/dev/null); do
  n=$(basename "${k:-shipped}" .so)
  export VH_LIB_PATH=${k:+$PWD/$k}
  [ -z "$k" ] && unset VH_LIB_PATH
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/$n -o p -- python3 bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extra-legs --preroll-seconds 0 > $O/$n.json 2> $O/$n.err
  python3 - $O/$n/p_counter_collection.csv $n <<'PY'
import csv,sys,collections
agg=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "k_render" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k:round(sum(v[len(v)//2:])/len(v[len(v)//2:])/1e6,3) for k,v in sorted(agg.items())})
PY
done
