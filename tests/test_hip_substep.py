"""The whole-substep kernel of the single-GPU step -- momentum and temperature wavefronts in one launch on the same
LDS tiles (k_momentum_dma<..., WITH_TEMP>) -- against the two separate kernels a latitude band keeps using
(DYCORE_STEP_FUSED=0): every field of the state and every diagnostic, stored or derived on demand, bit for bit,
on ragged grids (nx not a multiple of 64, ny not a multiple of 4), with moisture, in either precision; and the
alternate PVTFVB / PHI buffers it writes into are back where they were after a whole step."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.synthetic import initialize_fields
nx, ny, nz, steps, moist, f32, out, rad = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7], int(sys.argv[8])
dtype = np.float32 if f32 else np.float64
GR = Grid(Namelist(nz=nz, dlat_deg=180.0 / ny, dlon_deg=360.0 / nx, dt=1.0, i_moist_main_switch=moist, working_precision='float32' if f32 else 'float64'))
assert (GR.nx, GR.ny) == (nx, ny)
core = HipDycore(GR.nx, GR.ny, GR.nz, i_moist=moist, i_radiation=rad, dtype=dtype)
core.set_grid(GR.GRF['HIP'])
F = ModelFields(GR, core); initialize_fields(GR, F); F.copy_host_to_device()
if rad:      # a heating rate field as the radiation scheme would leave it (K / s, smooth, both signs)
    ii, jj, kk = np.meshgrid(np.arange(GR.nx + 2), np.arange(GR.ny + 2), np.arange(GR.nz), indexing='ij')
    core.upload('dPOTTdt_RAD', (2e-5 * np.sin(0.3 * ii + 0.2 * jj) * np.cos(0.5 * kk)).astype(dtype))
core.primary_diag()
ptr0 = {n: core.field_ptr(n)[0] for n in ('PHI', 'PVTFVB')}
core.step_matsuno(GR.dt, steps); core.sync()
ptr1 = {n: core.field_ptr(n)[0] for n in ('PHI', 'PVTFVB')}
names = ['COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'PHI', 'PVTFVB', 'POTTVB', 'PVTF', 'PHIVB'] + (['QV', 'QC'] if moist else [])
np.savez(out, **{n: core.download(n) for n in names})
st = core.kernel_stats() if hasattr(core, 'kernel_stats') else {}
print(json.dumps({'same_ptr': ptr0 == ptr1}))
''' % ROOT


def _run(tmp_path, tag, nx, ny, nz, steps, moist, f32, fused, rad=0):
    e = dict(os.environ, DYCORE_STEP_FUSED=str(fused))
    out = str(tmp_path / (tag + '.npz'))
    r = subprocess.run([sys.executable, '-c', WORKER, str(nx), str(ny), str(nz), str(steps), str(moist), str(f32), out, str(rad)],
                       env=e, check=True, capture_output=True, text=True)
    import json
    return np.load(out), json.loads(r.stdout.strip().splitlines()[-1])


@pytest.mark.parametrize('nx,ny,nz,steps,moist,f32,rad', [
    (90, 45, 8, 3, 0, 0, 0),         # BASELINE config 0's grid
    (130, 33, 16, 2, 0, 0, 0),       # ragged: three tile columns, the last one 2 wide; 33 rows = 8 tile rows + 1
    (64, 6, 8, 2, 0, 0, 0),          # one tile column, two tile rows
    (200, 50, 32, 2, 1, 0, 0),       # with moisture (round 3: advanced inside the fused launch; the separate run launches k_moist_canon)
    (200, 50, 32, 2, 1, 0, 1),       # BASELINE config 5's ingredients: moisture + a radiative heating rate (staged as one more own-cell tile)
    (360, 180, 64, 2, 0, 0, 0),      # BASELINE config 1's resolution at 64 levels
    (130, 33, 16, 2, 0, 1, 0),       # the float build (DYCORE_STEP_FUSED=1 asks for it; off by default there)
    (200, 50, 32, 2, 1, 1, 0),       # ... with the moisture sweep in its temperature wavefronts
])
def test_fused_substep_equals_separate_kernels(tmp_path, nx, ny, nz, steps, moist, f32, rad):
    ref, _ = _run(tmp_path, 'separate', nx, ny, nz, steps, moist, f32, 0, rad)
    got, info = _run(tmp_path, 'fused', nx, ny, nz, steps, moist, f32, 1, rad)
    for n in ref.files:
        assert np.array_equal(ref[n], got[n], equal_nan=True), n
    # two substeps swap the alternate buffers twice: pointers handed out before a step are still the fields' after it
    assert info['same_ptr']


@pytest.mark.skipif(any(k in os.environ for k in ('DYCORE_STEP_FUSED', 'DYCORE_LAZY_PHI', 'DYCORE_FUSED')),
                    reason='an A/B switch of the library is set in the environment')
def test_fused_substep_is_what_a_single_gpu_step_runs():
    from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
    from climate_model_amd.synthetic import initialize_fields
    GR = Grid(Namelist(nz=8, dlat_deg=4, dlon_deg=4))
    for kw, want in ((dict(), True), (dict(i_radiation=1), True), (dict(i_turb=1), False)):
        core = HipDycore(GR.nx, GR.ny, GR.nz, **kw)
        core.set_grid(GR.GRF['HIP'])
        F = ModelFields(GR, core); initialize_fields(GR, F); F.copy_host_to_device()
        core.primary_diag()
        core.profile(True)
        core.step_matsuno(GR.dt, 1); core.sync()
        st = {k: n for k, (n, t) in core.kernel_stats().items() if n}
        assert ('k_substep_fused' in st) == want, (kw, st)
        # (a turbulence-coupled handle: momentum kernel + the temperature kernel in the canonical forms, k_temp_canon<true>)
        assert ('k_temp_canon' in st) == (not want) and 'k_pott_fused' not in st, (kw, st)
        core.close()
