"""Whole-substep kernel (momentum + temperature wavefronts in one launch) against the two separate kernels:
bitwise comparison of the state after a few steps on several grids, then timing at 0.25 deg x 64.
usage: python tools/experiments/exp_substep_fused.py [--no-timing]"""
import os, subprocess, sys, json, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WORKER = r'''
import sys, os, json
sys.path.insert(0, %r)
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.synthetic import initialize_fields
d, nz, steps, out, moist = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d, i_moist_main_switch=moist))
core = HipDycore(GR.nx, GR.ny, GR.nz, i_moist=moist)
core.set_grid(GR.GRF['HIP'])
F = ModelFields(GR, core); initialize_fields(GR, F); F.copy_host_to_device()
core.primary_diag(); core.step_matsuno(GR.dt, steps); core.sync()
names = ['UWIND', 'VWIND', 'POTT', 'COLP', 'PHI', 'PVTFVB', 'POTTVB', 'PVTF', 'PHIVB'] + (['QV', 'QC'] if moist else [])
np.savez(out, **{n: core.download(n) for n in names})
if len(sys.argv) > 6:
    core.profile(True)
    ms = core.time_steps(GR.dt, 6)
    print(json.dumps({'ms': ms / 6, 'k': {k: t / n for k, (n, t) in core.kernel_stats().items() if n}}))
''' % ROOT

def run(d, nz, steps, fused, moist=0, timing=False):
    out = os.path.join(tempfile.gettempdir(), 'ssf_%d.npz' % fused)
    env = dict(os.environ, DYCORE_STEP_FUSED=str(fused))
    r = subprocess.run([sys.executable, '-c', WORKER, str(d), str(nz), str(steps), out, str(moist)] + (['t'] if timing else []),
                       env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print('FAILED', r.stderr[-2000:]); sys.exit(1)
    return np.load(out), (json.loads(r.stdout.strip().splitlines()[-1]) if timing else None)

bad = 0
for d, nz, steps, moist in ((10, 8, 3, 0), (4, 8, 5, 0), (3, 16, 3, 0), (1, 32, 2, 0), (2, 64, 2, 1), (0.5, 64, 2, 0)):
    a, _ = run(d, nz, steps, 0, moist)
    b, _ = run(d, nz, steps, 1, moist)
    worst = {}
    for n in a.files:
        x, y = a[n], b[n]
        m = np.isfinite(x) | np.isfinite(y)
        eq = np.array_equal(x[m], y[m])
        if not eq:
            den = np.nanmax(np.abs(x[m]))
            worst[n] = float(np.nanmax(np.abs(x[m] - y[m])) / den) if den > 0 else float('inf')
    print('d=%g nz=%d steps=%d moist=%d: %s' % (d, nz, steps, moist, 'bitwise equal' if not worst else 'DIFFERENT ' + json.dumps(worst)), flush=True)
    bad += bool(worst)
if '--no-timing' not in sys.argv:
    for fused in (0, 1, 0, 1):
        _, t = run(0.25, 64, 2, fused, 0, True)
        print('fused=%d  %.3f ms/step  %s' % (fused, t['ms'], '  '.join('%s=%.3f' % (k.replace('k_', ''), v) for k, v in t['k'].items() if v > 0.02)), flush=True)
sys.exit(1 if bad else 0)
