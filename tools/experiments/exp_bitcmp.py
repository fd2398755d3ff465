"""Bitwise comparison of two library builds (same ABI): N fused steps on a small grid, every
downloadable field of interest compared bit for bit.  usage: python tools/experiments/exp_bitcmp.py old.so new.so [dlat nz steps moist]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import sys, os
sys.path.insert(0, %r)
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.synthetic import initialize_fields, initialize_moisture
d, nz, steps, moist, out = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d))
core = HipDycore(GR.nx, GR.ny, GR.nz, i_moist=moist)
core.set_grid(GR.GRF['HIP'])
F = ModelFields(GR, core); initialize_fields(GR, F)
if moist: initialize_moisture(GR, F)
F.copy_host_to_device()
core.primary_diag(); core.step_matsuno(GR.dt, steps); core.sync()
names = ['COLP', 'UWIND', 'VWIND', 'POTT', 'PVTF', 'PVTFVB', 'PHI', 'PHIVB', 'POTTVB', 'WWIND'] + (['QV', 'QC'] if moist else [])
res = {n: core.download(n) for n in names}
core.secondary_diag()
for n in ('RHO', 'TAIR'): res[n] = core.download(n)
np.savez(out, **res)
''' % ROOT
a, b = sys.argv[1], sys.argv[2]
d, nz, steps, moist = (sys.argv[3:7] + ['4', '8', '3', '0'][len(sys.argv) - 3:]) if len(sys.argv) > 3 else ['4', '8', '3', '0']
import numpy as np
outs = []
for tag, lib in (('a', a), ('b', b)):
    env = dict(os.environ)
    if lib != 'default': env['DYCORE_LIB'] = os.path.join(ROOT, lib)
    out = '/tmp/bitcmp_%s.npz' % tag
    subprocess.run([sys.executable, '-c', WORKER, d, nz, steps, moist, out], env=env, check=True)
    outs.append(np.load(out))
for n in outs[0].files:
    x, y = outs[0][n], outs[1][n]
    same = np.array_equal(x, y, equal_nan=True)
    m = np.isfinite(x) & np.isfinite(y)
    err = np.abs(x[m] - y[m]).max() / max(np.abs(x[m]).max(), 1e-300) if m.any() else 0.0
    print('%-8s %s  max rel diff %.3e  nan-pattern equal %s' % (n, 'BITWISE' if same else 'DIFFERS', err, np.array_equal(np.isnan(x), np.isnan(y))))
