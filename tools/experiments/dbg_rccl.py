import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
import native_rccl_worker as W
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.bands import BandLayout, band_grid_arrays
from climate_model_amd.fields import FIELD_TABLE
from climate_model_amd.hip_backend import NativeBands, bands_unique_id
from climate_model_amd.synthetic import initialize_fields
torch.cuda.set_device(0)
stream = torch.cuda.Stream()
GR = Grid(Namelist(nz=8, dlat_deg=3, dlon_deg=3))
F = ModelFields(GR); initialize_fields(GR, F)
names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB']
L = BandLayout(GR.ny, 3)
grf = band_grid_arrays(GR.GRF['HIP'], L, 1)
state = {n: L.slice_rows(F.host[n], 1, FIELD_TABLE[n][1]) for n in names}
ny = L.ny(1)
def make(j_offset):
    c = HipDycore(GR.nx, ny, GR.nz, ny_global=GR.ny, j_offset=j_offset, stream=stream.cuda_stream)
    c.set_grid(grf)
    for n in names: c.upload(n, state[n])
    c.primary_diag()
    return c
check = ['COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'PVTFVB']
for nsteps in (1, 2):
    cores = [make(L.j_offset(r)) for r in range(3)]
    rb = NativeBands.local(cores); rb.step_matsuno(GR.dt, nsteps)
    ref = {n: cores[1].download(n) for n in check}
    rb.close(); [c.close() for c in cores]
    core = make(L.j_offset(1))
    nb = NativeBands.rccl(core, bands_unique_id(), 0, 1, 0, 0); nb.connect(); nb.step_matsuno(GR.dt, nsteps)
    for n in check:
        got = core.download(n)
        d = (got != ref[n]) & ~(np.isnan(got) & np.isnan(ref[n]))
        rows = np.where(d.any(axis=(0, 2)))[0]
        print('steps', nsteps, n, 'mismatch', int(d.sum()), 'rows', rows.tolist()[:30], 'max', float(np.nanmax(np.abs(np.where(d, got - ref[n], 0)))), flush=True)
    nb.close(); core.close()
