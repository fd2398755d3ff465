#!/bin/bash
# usage: tools/profile_run.sh <tag>  -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_bench.json
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_$1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$1 -o p --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $R/gpurun_out/$1_bench_profiled.json 2>/dev/null
cp $(find $R/gpurun_out/prof_$1 -name '*kernel_stats.csv' | head -1) $R/gpurun_out/$1_kernel_stats.csv
rm -rf $R/gpurun_out/prof_$1
cd $R && python3 bench.py > gpurun_out/$1_bench.json 2>/dev/null
tail -1 gpurun_out/$1_bench.json
