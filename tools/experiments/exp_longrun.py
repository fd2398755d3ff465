"""Stability probe: every scheme on (turbulence, moisture, radiation) through solver.run for a few simulated hours."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore, solver
from climate_model_amd.synthetic import initialize_fields, initialize_moisture
d, nz, nsteps = 2.0, 16, 3000
GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d, i_UVFLX_vert_turb=1, i_POTT_vert_turb=1, i_moist_main_switch=1,
                   i_radiation=1, rad_nth_hour=0.5))
core = HipDycore(GR.nx, GR.ny, GR.nz, i_turb=1, i_moist=1, i_radiation=1)
core.set_grid(GR.GRF['HIP'])
F = ModelFields(GR, core); initialize_fields(GR, F, noise_wind=0.0, noise_pott=0.0); initialize_moisture(GR, F)
F.host['SOILTEMP'][:, :, 0] = 295 - np.sin(GR.lat_rad[:, :, 0]) ** 2 * 30
F.host['SURFALBEDLW'][...] = 0.; F.host['SURFALBEDSW'][...] = 0.2
F.host['SSHFLX'][...] = 10.; F.host['SMOMXFLX'][...] = 0.; F.host['SMOMYFLX'][...] = 0.
F.host['dPOTTdt_RAD'][...] = 0.
F.copy_host_to_device()
t0 = time.perf_counter()
log = solver.run(GR, F, nsteps, nth_ts_print_diag=500)
core.sync()
print('%d steps of %g s (%.1f simulated hours) in %.1f s wall; GMT %s' % (nsteps, GR.dt, nsteps * GR.dt / 3600, time.perf_counter() - t0, GR.GMT))
for n in ('UWIND', 'POTT', 'QV', 'COLP', 'dPOTTdt_RAD'):
    a = core.download(n)[1:GR.nx + 1, 1:-1]
    print('  %-11s finite %s  min %.4g  max %.4g' % (n, bool(np.isfinite(a).all()), np.nanmin(a), np.nanmax(a)))
