"""Timing probe: the reference's loop body (secondary diagnostics, turbulence scheme, Matsuno step with
moisture and the turbulent terms) at the bench grid through solver.run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore, solver
from climate_model_amd.synthetic import initialize_fields, initialize_moisture
d = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25
GR = Grid(Namelist(nz=64, dlat_deg=d, dlon_deg=d, i_UVFLX_vert_turb=1, i_POTT_vert_turb=1, i_moist_main_switch=1))
core = HipDycore(GR.nx, GR.ny, GR.nz, i_turb=1, i_moist=1)
core.set_grid(GR.GRF['HIP'])
F = ModelFields(GR, core); initialize_fields(GR, F); initialize_moisture(GR, F)
F.host['SSHFLX'][...] = 20.; F.host['SMOMXFLX'][...] = 0.01; F.host['SMOMYFLX'][...] = 0.01
F.copy_host_to_device()
solver.run(GR, F, 2, nth_ts_print_diag=1000)
core.sync(); core.profile(1)
t0 = time.perf_counter(); n = 5
solver.run(GR, F, n, nth_ts_print_diag=1000)
core.sync(); ms = (time.perf_counter() - t0) / n * 1e3
st = {k: round(v[1] / n, 3) for k, v in core.kernel_stats().items() if v[0]}
print('%g deg x 64, turbulence + moisture: %.2f ms per loop step = %.2f G cell-updates/s  %s' % (d, ms, GR.nx * GR.ny * GR.nz / ms / 1e6, st))
print('finite:', bool(np.isfinite(core.download('UWIND')[1:-2, 1:-1]).all()))
