#!/bin/bash
# A/B of the momentum kernel's work list (MomPlan) at the bench grid: bench.py per setting, one line each
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" python bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readlines()[-1]); r=j['roofline']
print('ms/step %.3f  mom avg %.4f ms  frac %.3f  %s' % (j['ms_per_step'], r['avg_launch_ms'], r['frac'], {k:round(v,3) for k,v in r['kernels_ms_per_step'].items()}))"; }
run DYCORE_MOM_SPLIT=0 DYCORE_MOM_KS=1
run DYCORE_MOM_SPLIT=0 DYCORE_MOM_KS=2
run DYCORE_MOM_SPLIT=0 DYCORE_MOM_KS=4
run DYCORE_MOM_SPLIT=1
run DYCORE_MOM_SPLIT=1 DYCORE_MOM_PRO=6
run DYCORE_MOM_SPLIT=1
