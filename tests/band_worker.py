"""Worker of test_two_process_gloo_bands_equal_single (one rank = one latitude band)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import numpy as np
import torch
import torch.distributed as dist

from climate_model_amd import HipDycore
from climate_model_amd.bands import BandLayout, DistributedBands, band_grid_arrays, gather_global
from climate_model_amd.fields import FIELD_TABLE
from test_hip_bands import _initial_state


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    backend = os.environ.get('DYCORE_BANDS_BACKEND', 'gloo')
    dev = 0 if os.environ.get('DYCORE_BANDS_ONE_GPU') else int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(dev)
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', dev))
    else:
        dist.init_process_group(backend)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    GR, state = _initial_state(4, 8, 0)
    L = BandLayout(GR.ny, world)
    core = HipDycore(GR.nx, L.ny(rank), GR.nz, ny_global=GR.ny, j_offset=L.j_offset(rank), device=dev,
                     stream=stream.cuda_stream)
    core.set_grid(band_grid_arrays(GR.GRF['HIP'], L, rank))
    for n, a in state.items():
        core.upload(n, L.slice_rows(a, rank, FIELD_TABLE[n][1]))
    core.primary_diag()
    bands = DistributedBands(core, L, rank, torch.device('cuda', dev))
    if os.environ.get('DYCORE_EXPECT_NATIVE'):      # distinct devices + nccl: the in-library RCCL transport must be the one that runs
        assert bands.native is not None, 'in-library transport not used: %s' % bands.native_error
    bands.connect()             # as bench.py does: must leave every field untouched
    bands.step_matsuno(GR.dt, 3)
    core.sync()
    out = {}
    for n in ('COLP', 'UWIND', 'VWIND', 'POTT'):
        parts = [None] * world
        dist.all_gather_object(parts, core.download(n))
        out[n] = gather_global(L, parts, n)
    if rank == 0:
        np.savez(sys.argv[1], **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
