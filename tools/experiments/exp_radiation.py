"""Timing probe: one radiation call at the bench grid (0.25 deg x 64), HIP events."""
import datetime, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore, Radiation
from climate_model_amd.synthetic import initialize_fields, initialize_moisture

d, nz = (float(sys.argv[1]) if len(sys.argv) > 1 else 0.25), 64
GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d, i_radiation=1))
core = HipDycore(GR.nx, GR.ny, GR.nz, i_radiation=1, i_moist=1)
core.set_grid(GR.GRF['HIP'])
F = ModelFields(GR, core)
initialize_fields(GR, F); initialize_moisture(GR, F)
F.host['SOILTEMP'][:, :, 0] = 295 - np.sin(GR.lat_rad[:, :, 0]) ** 2 * 30
F.host['SURFALBEDLW'][...] = 0.; F.host['SURFALBEDSW'][...] = 0.2
F.copy_host_to_device()
core.primary_diag(); core.secondary_diag()
RAD = Radiation(GR)
gmt = datetime.datetime(2019, 6, 21, 14, 0, 0)
args = F.get(RAD.fields_radiation, target='HIP')
RAD.simple_radiation(GR, host=F.host, current_GMT=gmt, **args)
core.sync()
core.profile(1)
t0 = time.perf_counter()
for _ in range(3):
    RAD.simple_radiation(GR, host=F.host, current_GMT=gmt, **args)
core.sync()
wall = (time.perf_counter() - t0) / 3
st = core.kernel_stats()
n, ms = st['k_radiation']
print('grid %dx%dx%d: k_radiation %.2f ms per call (device), %.1f ms wall incl. host astronomy + uploads; %d columns, %.2f us/column' % (
    GR.nx, GR.ny, GR.nz, ms / n, wall * 1e3, GR.nx * GR.ny, ms / n * 1e3 / (GR.nx * GR.ny)))
heat = core.download('dPOTTdt_RAD')[1:-1, 1:-1]
print('dPOTTdt_RAD range [%.3e, %.3e] K/s, finite: %s' % (heat.min(), heat.max(), np.isfinite(heat).all()))
