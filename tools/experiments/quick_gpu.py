"""Ad-hoc first GPU check: timing of the staged v1 kernels at a few grid sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.synthetic import initialize_fields

for d, nz, steps in [(1, 32, 20), (0.5, 64, 10), (0.25, 64, 10)]:
    t = time.time()
    GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d))
    core = HipDycore(GR.nx, GR.ny, GR.nz)
    core.set_grid(GR.GRF['HIP'])
    F = ModelFields(GR, core)
    initialize_fields(GR, F)
    F.copy_host_to_device()
    core.primary_diag()
    core.sync()
    print('setup %.1fs' % (time.time() - t), flush=True)
    core.step_matsuno(GR.dt, 2)
    core.sync()
    core.profile(True)
    ms = core.time_steps(GR.dt, steps)
    st = core.kernel_stats()
    cells = GR.nx * GR.ny * GR.nz
    per = ms / steps
    print('grid %gdeg x%d  cells %d  %.3f ms/step  %.3f Gcell-upd/s  alg-BW %.1f GB/s (%.1f%% of 8TB/s)' % (
        d, nz, cells, per, cells / per / 1e6, cells * 1040 / per / 1e6, cells * 1040 / per / 1e6 / 80), flush=True)
    for k, (n, tot) in st.items():
        if n:
            print('   %-18s %5d launches  %9.3f ms avg  %5.1f%%' % (k, n, tot / n, 100 * tot / ms))
    u = core.download('UWIND')
    print('   finite:', np.isfinite(u[1:-2, 1:-1]).all(), 'max|U|', np.abs(u[1:-2, 1:-1]).max(), flush=True)
    core.close()
    del F, GR
