#!/bin/bash
# End-of-round evidence on the GPU box: kernel stats + three PMC passes of the default bench command, and the bench
# lines (default, offline, cfg3, cfg4), into gpurun_out/final/.  Summarise with tools/summarize_profile.py.
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o p -- python3 bench.py --no-cpu-baseline > $O/trace.log 2>&1
grep -h '"metric"' $O/trace.log > $O/bench_under_profiler.json
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$N -o p -- python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline > $O/pmc_$N.log 2>&1
done
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 python3 bench.py --offline --no-cpu-baseline > $O/bench_offline.json 2> $O/bench_offline.err
timeout -k 10 300 python3 bench.py --config cfg3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err
timeout -k 10 300 python3 bench.py --config cfg4 --no-cpu-baseline > $O/bench_cfg4.json 2> $O/bench_cfg4.err
ls $O
