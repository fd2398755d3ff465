"""Compute-side strong-scaling probe on ONE GPU: run the N bands of the bench grid one after
another (LocalBands) and report the mean per-band kernel time per step = what each of N GPUs
would spend in kernels (exchange excluded)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.bands import BandLayout, LocalBands, band_grid_arrays
from climate_model_amd.fields import FIELD_TABLE
from climate_model_amd.synthetic import initialize_fields

d, nz, steps = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25, 64, 4
GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d))
F = ModelFields(GR); initialize_fields(GR, F)
names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB']
base = None
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)     # one stream for every band: kernels run one after another
for nb in [1, 2, 4, 8]:
    L = BandLayout(GR.ny, nb)
    cores = []
    for r in range(nb):
        c = HipDycore(GR.nx, L.ny(r), GR.nz, ny_global=GR.ny, j_offset=L.j_offset(r), stream=stream.cuda_stream)
        c.set_grid(band_grid_arrays(GR.GRF['HIP'], L, r))
        for n in names:
            c.upload(n, L.slice_rows(F.host[n], r, FIELD_TABLE[n][1]))
        c.primary_diag()
        cores.append(c)
    lb = LocalBands(cores, L, torch.device('cuda:0'), overlap=bool(int(os.environ.get('OVERLAP', '1'))), native=bool(int(os.environ.get('NATIVE', '0'))))
    lb.step_matsuno(GR.dt, 1)
    for c in cores: c.profile(True)
    lb.step_matsuno(GR.dt, steps)
    tot = {}
    for c in cores:
        for k, (n, ms) in c.kernel_stats().items():
            if n: tot[k] = tot.get(k, 0.) + ms
    for c in cores: c.profile(False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lb.step_matsuno(GR.dt, steps)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / steps / nb * 1e3
    for c in cores: c.profile(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    lb.step_matsuno(GR.dt, steps)
    torch.cuda.synchronize(); wall_prof = (time.perf_counter() - t0) / steps / nb * 1e3
    for c in cores: c.profile(False)
    per_gpu = sum(tot.values()) / nb / steps
    if base is None: base = per_gpu
    print('bands=%d  wall %.3f (profiled %.3f) ms/band-step  kernel ms/step per band: %.3f   compute-only speedup %.2fx (eff %.0f%%)  | %s' % (
        nb, wall, wall_prof, per_gpu, base / per_gpu, 100 * base / per_gpu / nb,
        '  '.join('%s=%.3f' % (k.replace('k_', ''), v / nb / steps) for k, v in tot.items() if v / nb / steps > 0.005)), flush=True)
    for c in cores: c.close()
