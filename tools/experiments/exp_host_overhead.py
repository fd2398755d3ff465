"""Host-side cost of the overlapped band schedule: 2 bands of a tiny grid in one process, so that the
GPU work is negligible and the wall time per step is Python + ctypes + torch stream/event calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.bands import BandLayout, LocalBands, band_grid_arrays
from climate_model_amd.fields import FIELD_TABLE
from climate_model_amd.synthetic import initialize_fields

GR = Grid(Namelist(nz=8, dlat_deg=4, dlon_deg=4))
F = ModelFields(GR); initialize_fields(GR, F)
names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB']
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
for nb in (1, 2):
    L = BandLayout(GR.ny, nb)
    cores = []
    for r in range(nb):
        c = HipDycore(GR.nx, L.ny(r), GR.nz, ny_global=GR.ny, j_offset=L.j_offset(r), stream=stream.cuda_stream)
        c.set_grid(band_grid_arrays(GR.GRF['HIP'], L, r))
        for n in names:
            c.upload(n, L.slice_rows(F.host[n], r, FIELD_TABLE[n][1]))
        c.primary_diag(); cores.append(c)
    lb = LocalBands(cores, L, torch.device('cuda:0'), native=bool(int(os.environ.get('NATIVE', '0'))))
    lb.step_matsuno(GR.dt, 5)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 200
    lb.step_matsuno(GR.dt, n)
    t_issue = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); t_all = (time.perf_counter() - t0) / n
    print('bands=%d: host issue time %.3f ms/step, wall %.3f ms/step' % (nb, t_issue * 1e3, t_all * 1e3))
    for c in cores: c.close()
