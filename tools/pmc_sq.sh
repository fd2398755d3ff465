#!/bin/bash
# usage: tools/pmc_sq.sh <tag>  -> per-kernel SQ counters (wave cycles, wait, active VALU/any) of a short bench run
set -e
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_sq_$1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $R/gpurun_out/pmc_sq_$1 -o s --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in glob.glob('$R/gpurun_out/pmc_sq_$1/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        tot[k][row['Counter_Name']] += float(row['Counter_Value']); n[k].add(row['Dispatch_Id'])
for k in sorted(tot):
    if 'momentum' in k or 'substep' in k or 'pott' in k or 'continuity_reg' in k or 'moist' in k:
        t = tot[k]; nl = len(n[k]); wc = t['SQ_WAVE_CYCLES'] or 1
        print('%-34s n=%2d waves %.0f  wave_cycles %.3e  wait_any %.0f%%  wait_inst %.0f%%  active_any %.0f%%  active_valu %.0f%%  valu insts/wave %.0f' % (
            k[:34], nl, t['SQ_WAVES'] / nl, wc / nl, 100 * t['SQ_WAIT_ANY'] / wc, 100 * t['SQ_WAIT_INST_ANY'] / wc,
            100 * t['SQ_ACTIVE_INST_ANY'] / wc, 100 * t['SQ_ACTIVE_INST_VALU'] / wc, t['SQ_INSTS_VALU'] / max(t['SQ_WAVES'], 1)))
PY
rm -rf $R/gpurun_out/pmc_sq_$1
