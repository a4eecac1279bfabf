#!/bin/bash
# PMC passes over the dense-scene integrate kernel (separate runs, no trace domains)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=${1:-gpurun_out/pmc_dense}
CFG=${2:---config cfg3 --scene S2 --no-streaming}
mkdir -p $O
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
         "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $O/p$i -o p -- python3 bench.py $CFG --steps 12 --warmup 4 --no-cpu-baseline --no-extra-legs --preroll-seconds 0 > $O/p$i.json 2> $O/p$i.err; echo "pmc pass $i rc=$?"
done
python3 - "$O" <<'PY'
import csv,glob,sys,collections
O=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(O+"/p*/p_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        for name in ("k_integrate_fused","k_render","k_interval_splat","k_compute_normals","k_alloc","k_compactify"):
            if name in k:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(O+"/summary.txt","w") as out:
    for k,v in agg.items():
        out.write(k+"\n")
        for c,vals in sorted(v.items()):
            tail=vals[len(vals)//2:]
            out.write(f"   {c:40s} n={len(vals):4d} avg(last half)={sum(tail)/len(tail):16.1f}\n")
print(open(O+"/summary.txt").read())
PY
