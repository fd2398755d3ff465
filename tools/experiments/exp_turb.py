"""Timing probe: turbulence-coupled Matsuno step at the bench grid, fused (default) vs stage kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.synthetic import initialize_fields
d, nz = (float(sys.argv[1]) if len(sys.argv) > 1 else 0.25), 64
GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d, i_UVFLX_vert_turb=1, i_POTT_vert_turb=1))
F = ModelFields(GR); initialize_fields(GR, F)
rng = np.random.default_rng(1)
for n, lo, hi in [('KMOM', 1e-6, 0.01), ('KHEAT', 1e-6, 0.01), ('SSHFLX', -50., 150.), ('SMOMXFLX', -0.2, 0.2), ('SMOMYFLX', -0.2, 0.2)]:
    F.host[n][...] = rng.uniform(lo, hi, F.host[n].shape)
names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB', 'KMOM', 'KHEAT', 'SSHFLX', 'SMOMXFLX', 'SMOMYFLX']
for fused in ('1', '0'):
    os.environ['DYCORE_FUSED_TURB'] = fused
    core = HipDycore(GR.nx, GR.ny, GR.nz, i_turb=1)
    core.set_grid(GR.GRF['HIP'])
    for n in names:
        core.upload(n, F.host[n])
    core.primary_diag(); core.step_matsuno(GR.dt, 2); core.sync()
    core.profile(1)
    t0 = time.perf_counter(); core.step_matsuno(GR.dt, 5); core.sync()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    st = {k: round(v[1] / 5, 3) for k, v in core.kernel_stats().items() if v[0]}
    print('DYCORE_FUSED_TURB=%s: %.2f ms per step  %s' % (fused, ms, st))
    u = core.download('UWIND'); print('   finite:', bool(np.isfinite(u[1:-2, 1:-1]).all()))
    core.close()
