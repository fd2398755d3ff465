"""The momentum kernel's work list (MomPlan: tile rows x level chunks, two segments) never changes a bit
of the result: every forced number of level chunks per tile, and the two-segment plan switched off, give the
same fields as the default plan -- chunks that start inside a column restart the vertical recurrences from
the stored WWIND and winds."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import sys
sys.path.insert(0, %r)
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.synthetic import initialize_fields
nx, ny, nz, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
GR = Grid(Namelist(nz=nz, dlat_deg=180.0 / ny, dlon_deg=360.0 / nx, dt=1.0))
assert (GR.nx, GR.ny) == (nx, ny)
core = HipDycore(GR.nx, GR.ny, GR.nz)
core.set_grid(GR.GRF['HIP'])
F = ModelFields(GR, core); initialize_fields(GR, F); F.copy_host_to_device()
core.primary_diag(); core.step_matsuno(GR.dt, 2); core.sync()
np.savez(out, **{n: core.download(n) for n in ('COLP', 'UWIND', 'VWIND', 'POTT')})
''' % ROOT


def _run(tmp_path, tag, nx, ny, nz, **env):
    e = dict(os.environ)
    e.update({k: str(v) for k, v in env.items()})
    out = str(tmp_path / (tag + '.npz'))
    subprocess.run([sys.executable, '-c', WORKER, str(nx), str(ny), str(nz), out], env=e, check=True)
    return np.load(out)


@pytest.mark.gpu
@pytest.mark.parametrize('nx,ny,nz', [(720, 360, 13), (90, 45, 8), (130, 33, 21)])
def test_level_chunks_and_segments_are_bitwise_neutral(tmp_path, nx, ny, nz):
    ref = _run(tmp_path, 'default', nx, ny, nz)
    variants = [dict(DYCORE_MOM_SPLIT=0, DYCORE_MOM_KS=1), dict(DYCORE_MOM_SPLIT=0)]
    variants += [dict(DYCORE_MOM_KS=ks) for ks in (2, 3, 5) if nz // ks >= 1]
    for i, env in enumerate(variants):
        got = _run(tmp_path, 'v%d' % i, nx, ny, nz, **env)
        for n in ref.files:
            assert np.array_equal(ref[n], got[n], equal_nan=True), (env, n)
