"""The in-library band step over RCCL on ONE GPU (two ranks cannot share a GPU under RCCL): one rank whose band is
its own southern and northern neighbour, so that every message really goes through ncclSend / ncclRecv on the
library's comm stream.  Reference: the same schedule with device copies (NativeBands.local) on three bands that all
hold this band's fields and grid constants -- the middle one then receives exactly the rows it would send itself.
(The outer two touch a pole; within the two steps run here that cannot reach their far edge, 20 rows away.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch

from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.bands import BandLayout, band_grid_arrays
from climate_model_amd.fields import FIELD_TABLE
from climate_model_amd.hip_backend import NativeBands, bands_unique_id
from climate_model_amd.synthetic import initialize_fields, initialize_moisture


def main():
    i_moist = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    dtype = np.float32 if (len(sys.argv) > 2 and sys.argv[2] == 'f32') else np.float64       # the build (working precision)
    torch.cuda.set_device(0)
    stream = torch.cuda.Stream()
    GR = Grid(Namelist(nz=8, dlat_deg=3, dlon_deg=3, i_moist_main_switch=i_moist))      # 120 x 60
    F = ModelFields(GR)
    initialize_fields(GR, F)
    names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB']
    if i_moist:
        initialize_moisture(GR, F)
        names += ['QV', 'QC', 'SLHFLX']
    L = BandLayout(GR.ny, 3)
    grf = band_grid_arrays(GR.GRF['HIP'], L, 1)
    state = {n: L.slice_rows(F.host[n], 1, FIELD_TABLE[n][1]) for n in names}
    ny = L.ny(1)

    def make(j_offset):
        c = HipDycore(GR.nx, ny, GR.nz, ny_global=GR.ny, j_offset=j_offset, i_moist=i_moist, stream=stream.cuda_stream, dtype=dtype)
        c.set_grid(grf)
        for n in names:
            c.upload(n, state[n])
        c.primary_diag()
        return c

    check = ['COLP', 'UWIND', 'VWIND', 'POTT', 'PHI', 'WWIND'] + (['QV', 'QC'] if i_moist else [])
    cores = [make(L.j_offset(r)) for r in range(3)]
    ref_bands = NativeBands.local(cores)
    ref_bands.step_matsuno(GR.dt, 2)
    ref = {n: cores[1].download(n) for n in check}
    ref_bands.close()
    for c in cores:
        c.close()

    core = make(L.j_offset(1))
    nb = NativeBands.rccl(core, bands_unique_id(dtype), 0, 1, 0, 0)
    nb.connect()
    nb.step_matsuno(GR.dt, 2)
    for n in check:
        got = core.download(n)
        np.testing.assert_array_equal(got, ref[n], err_msg=n)
        assert np.isfinite(got[1:GR.nx + 1, 1:ny + 1]).all(), n
    assert np.abs(ref['UWIND'][1:-2, 1:-1] - state['UWIND'][1:-2, 1:-1]).max() > 0
    nb.close()
    core.close()
    print('NATIVE_RCCL_OK')


if __name__ == '__main__':
    main()
