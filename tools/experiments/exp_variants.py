"""A/B timing of library builds (same ABI) on the bench workload.  usage:
   python tools/experiments/exp_variants.py [--d 0.25] [--nz 64] [--steps 6] lib1.so lib2.so ...
Each library runs in its own subprocess (DYCORE_LIB); rounds are interleaved."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORKER = r'''
import sys, os, json
sys.path.insert(0, %r)
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.synthetic import initialize_fields
d, nz, steps, rounds = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d))
core = HipDycore(GR.nx, GR.ny, GR.nz)
core.set_grid(GR.GRF['HIP'])
F = ModelFields(GR, core); initialize_fields(GR, F); F.copy_host_to_device()
core.primary_diag(); core.step_matsuno(GR.dt, 2); core.sync()
res = []
for r in range(rounds):
    core.profile(True)
    ms = core.time_steps(GR.dt, steps)
    st = core.kernel_stats()
    res.append({'ms_per_step': ms / steps, 'kernels': {k: t / n for k, (n, t) in st.items() if n}})
u = core.download('UWIND')
print(json.dumps({'res': res, 'finite': bool(np.isfinite(u[1:-2, 1:-1]).all()), 'chk': float(np.abs(u[1:-2,1:-1]).sum())}))
''' % ROOT

def main():
    args = sys.argv[1:]
    d, nz, steps, rounds, reps = 0.25, 64, 6, 3, 1
    libs = []
    i = 0
    while i < len(args):
        if args[i] == '--d': d = float(args[i+1]); i += 2
        elif args[i] == '--nz': nz = int(args[i+1]); i += 2
        elif args[i] == '--steps': steps = int(args[i+1]); i += 2
        elif args[i] == '--rounds': rounds = int(args[i+1]); i += 2
        elif args[i] == '--reps': reps = int(args[i+1]); i += 2        # the whole list again, in the same order (A B C A B C): drift between processes shows
        else: libs.append(args[i]); i += 1
    for lib in libs * reps:
        env = dict(os.environ)
        name = lib
        if '=' in lib:                      # NAME=path[,ENV=VAL...]
            name, rest = lib.split('=', 1)
            parts = rest.split(',')
            lib = parts[0]
            for kv in parts[1:]:
                k, v = kv.split(':', 1); env[k] = v
        if lib and lib != 'default':
            env['DYCORE_LIB'] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, '-c', WORKER, str(d), str(nz), str(steps), str(rounds)],
                           env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(name, 'FAILED', r.stderr[-1500:]); continue
        j = json.loads(r.stdout.strip().splitlines()[-1])
        best = min(j['res'], key=lambda x: x['ms_per_step'])
        ks = '  '.join('%s=%.3f' % (k.replace('k_', ''), v) for k, v in best['kernels'].items() if v > 0.02)
        print('%-28s %.3f ms/step (rounds %s) finite=%s chk=%.6e\n      %s' % (
            name, best['ms_per_step'], ' '.join('%.3f' % x['ms_per_step'] for x in j['res']), j['finite'], j['chk'], ks), flush=True)

main()
