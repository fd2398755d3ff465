"""Long run of the dry / moist + radiation step through solver.run on one GPU: the whole-substep kernel against the two
separate kernels, bit for bit after thousands of steps (status lines and final state)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore, solver
from climate_model_amd.synthetic import initialize_fields, initialize_moisture
d, nz, nsteps, out = float(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d, i_moist_main_switch=1, i_radiation=1, rad_nth_hour=0.5))
core = HipDycore(GR.nx, GR.ny, GR.nz, i_moist=1, i_radiation=1)
core.set_grid(GR.GRF['HIP'])
F = ModelFields(GR, core); initialize_fields(GR, F); initialize_moisture(GR, F)
F.host['SOILTEMP'][:, :, 0] = 295 - np.sin(GR.lat_rad[:, :, 0]) ** 2 * 30
F.host['SURFALBEDLW'][...] = 0.; F.host['SURFALBEDSW'][...] = 0.2
F.host['dPOTTdt_RAD'][...] = 0.
F.copy_host_to_device()
lines = []
solver.run(GR, F, nsteps, nth_ts_print_diag=500, out=lines.append)
core.sync()
np.savez(out, lines=np.array(lines), **{n: core.download(n) for n in ('UWIND', 'VWIND', 'POTT', 'COLP', 'QV', 'QC', 'PHI', 'dPOTTdt_RAD')})
print(lines[-1])
''' % ROOT
res = {}
for fused in (0, 1):
    out = os.path.join(tempfile.gettempdir(), 'lr_%d.npz' % fused)
    r = subprocess.run([sys.executable, '-c', WORKER, '2.0', '16', '4000', out], env=dict(os.environ, DYCORE_STEP_FUSED=str(fused)), capture_output=True, text=True)
    if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
    print('fused=%d  %s' % (fused, r.stdout.strip().splitlines()[-1]))
    res[fused] = np.load(out)
bad = [n for n in res[0].files if not np.array_equal(res[0][n], res[1][n], equal_nan=(n != 'lines'))] if True else []
u = res[1]['UWIND'][1:-2, 1:-1]
print('finite:', bool(np.isfinite(u).all()), ' max |U| %.2f' % np.abs(u).max(), ' bitwise equal after 4000 steps:', not bad, bad)
sys.exit(1 if bad else 0)
