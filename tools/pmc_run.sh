#!/bin/bash
# usage: tools/pmc_run.sh <tag>   (env switches are inherited)  -> gpurun_out/pmc_<tag>.json
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_f_$1 $R/gpurun_out/pmc_w_$1
rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f_$1 -o f --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w_$1 -o w --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_f_$1 $R/gpurun_out/pmc_w_$1 $R/gpurun_out/pmc_$1.json
rm -rf $R/gpurun_out/pmc_f_$1 $R/gpurun_out/pmc_w_$1
