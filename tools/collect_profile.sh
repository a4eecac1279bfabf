#!/bin/bash
# End-of-round evidence on the GPU box, into gpurun_out/final/ (summarise with tools/summarize_profile.py):
#   kernel stats + PMC passes of the default bench command and of the dense-scene run, and the bench lines.
# Counters are collected in passes of their own (--pmc only, no trace domains).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
B="--no-cpu-baseline --no-extra-legs"
D="--config cfg3 --scene S2 --no-streaming --steps 100 --warmup 20 $B"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 bench.py $B > $O/trace.log 2>&1; echo "trace rc=$?"
grep -h '"metric"' $O/trace.log > $O/bench_under_profiler.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dense -o p -- python3 bench.py $D > $O/trace_dense.log 2>&1; echo "trace dense rc=$?"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-24)
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$N -o p -- python3 bench.py --steps 100 --warmup 20 $B > $O/pmc_$N.log 2>&1; echo "pmc $N rc=$?"
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/pmcdense_$N -o p -- python3 bench.py $D > $O/pmcdense_$N.log 2>&1; echo "pmc dense $N rc=$?"
done
timeout -k 10 500 python3 bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_20_5.json 2> $O/bench_driver_20_5.err; echo "bench 20/5 rc=$?"
timeout -k 10 300 python3 bench.py --offline $B > $O/bench_offline.json 2> $O/bench_offline.err
timeout -k 10 300 python3 bench.py --python-loop --no-alloc-ahead $B > $O/bench_python_loop.json 2> $O/bench_python_loop.err
timeout -k 10 300 python3 bench.py --no-alloc-ahead $B > $O/bench_no_riders.json 2> $O/bench_no_riders.err
timeout -k 10 300 python3 bench.py --frames-on-host $B > $O/bench_frames_on_host.json 2> $O/bench_frames_on_host.err
timeout -k 10 300 python3 bench.py --config cfg3 $B > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout -k 10 300 python3 bench.py --config cfg3 --no-streaming $B > $O/bench_cfg3_no_streaming.json 2> $O/bench_cfg3_no_streaming.err
timeout -k 10 300 python3 bench.py --config cfg3 --streaming-radius 1.2 --streaming-pos-z 1.6 --streaming-extent 0.5 $B > $O/bench_cfg3_streaming_traffic.json 2> $O/bench_cfg3_streaming_traffic.err
timeout -k 10 300 python3 bench.py --config cfg4 $B > $O/bench_cfg4.json 2> $O/bench_cfg4.err
timeout -k 10 300 python3 bench.py --config cfg3 --scene S2 --no-streaming $B > $O/bench_dense_s2.json 2> $O/bench_dense_s2.err
timeout -k 10 300 python3 bench.py --config cfg3 --streaming-radius 1.2 --streaming-pos-z 1.6 --streaming-extent 0.5 --steps 300 --warmup 100 $B > $O/bench_cfg3_streaming_traffic_300.json 2> $O/bench_cfg3_streaming_traffic_300.err
timeout -k 10 200 python3 tools/valu_issue_probe.py --out $O/valu_issue.json > $O/valu_issue.log 2>&1
ls $O | head -60
