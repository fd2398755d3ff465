#!/bin/bash
# usage (on the GPU box): bash tools/artefacts.sh <tag>   -> the measurement set of a build under gpurun_out/<tag>_*
# (bench + rocprofv3 kernel stats, PMC traffic, SQ counters, other grids, config 5, turbulence-coupled step, band emulation)
set -e
t=$1
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_run.sh $t
bash tools/pmc_run.sh $t > gpurun_out/${t}_pmc.log 2>&1 && cp gpurun_out/pmc_$t.json gpurun_out/${t}_pmc_traffic.json
bash tools/pmc_sq.sh $t > gpurun_out/${t}_sq.txt 2>&1
python3 bench.py --dlat 1 --nz 32 --no-cpu-baseline > gpurun_out/${t}_bench_1deg.json 2>/dev/null
python3 bench.py --dlat 0.5 --nz 64 --no-cpu-baseline > gpurun_out/${t}_bench_05deg.json 2>/dev/null
python3 bench.py --dlat 4 --nz 8 --no-cpu-baseline > gpurun_out/${t}_bench_4deg.json 2>/dev/null
python3 bench.py --moist --radiation --no-cpu-baseline > gpurun_out/${t}_bench_cfg5.json 2>/dev/null
python3 bench.py --f32 --no-cpu-baseline > gpurun_out/${t}_bench_f32.json 2>/dev/null
python3 tools/experiments/exp_turb.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${t}_turb.txt
NATIVE=1 python3 tools/experiments/exp_bands_local.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${t}_bands_native.txt
echo artefacts $t done
