#!/bin/bash
# usage: tools/pmc_counters.sh <tag> "<COUNTER ...>" ["<COUNTER ...>" ...]   one rocprofv3 --pmc pass per quoted group over a short bench run;
# prints per kernel (continuity / momentum / substep / temp / moist / turb) the counter sums per launch.  Env switches (DYCORE_LIB ...) are inherited.
# PMC_PROG="tools/experiments/exp_turb.py 0.25": another program (path relative to the repo root + arguments) instead of the bench run.
set -e
tag=$1; shift
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
n=0
for grp in "$@"; do
  n=$((n+1))
  rm -rf $R/gpurun_out/pmc_c_${tag}_$n
  rocprofv3 --pmc $grp -d $R/gpurun_out/pmc_c_${tag}_$n -o s --output-format csv -- python3 $R/${PMC_PROG:-bench.py --steps 3 --warmup 1 --no-cpu-baseline} > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob('$R/gpurun_out/pmc_c_${tag}_*/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'].split('(')[0].replace('void ', '')
        tot[k][row['Counter_Name']] += float(row['Counter_Value']); n[k][row['Counter_Name']].add(row['Dispatch_Id'])
for k in sorted(tot):
    if any(s in k for s in ('momentum', 'substep', 'temp_canon', 'pott', 'continuity_reg', 'moist', 'turb')):
        print(k[:90])
        for c in sorted(tot[k]):
            print('    %-34s %14.4g per launch' % (c, tot[k][c] / max(1, len(n[k][c]))))
PY
rm -rf $R/gpurun_out/pmc_c_${tag}_*
