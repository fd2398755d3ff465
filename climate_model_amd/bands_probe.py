"""First contact of the in-library RCCL band transport, run as a CHILD of each rank before the rank itself commits
to it (bands._negotiate_native):

    python -m climate_model_amd.bands_probe <id hex> <rank> <nranks> <south> <north> <device> <dtype>

One small latitude band per rank on the rank's own device: dycore_bands_create_rccl (ncclCommInitRank next to whatever
communicators the parent job holds), dycore_bands_connect (the peer connections over xGMI), two Matsuno steps through
ncclSend / ncclRecv, a finite-value check.  Prints BANDS_PROBE_OK and exits 0; anything else -- an error, a crash, or a
hang that the parent ends at its timeout -- makes every rank of the job fall back to the torch.distributed transport
together.  The probe grid has 6 rows per band; its results are not used for anything."""
import sys

import numpy as np


def main(argv):
    unique_id = bytes.fromhex(argv[0])
    rank, nranks, south, north, device = (int(a) for a in argv[1:6])
    dtype = np.dtype(argv[6]) if len(argv) > 6 else np.dtype(np.float64)
    from . import Namelist, Grid, ModelFields, HipDycore
    from .bands import BandLayout, band_grid_arrays
    from .fields import FIELD_TABLE
    from .hip_backend import NativeBands
    from .synthetic import initialize_fields

    # 6 rows per band, 12 longitudes per band.  A single rank (the one-GPU test of this module) is its own neighbour on
    # both sides: it takes the middle one of three bands
    nb_grid, band = (nranks, rank) if nranks > 1 else (3, 1)
    d = 30.0 / nb_grid
    GR = Grid(Namelist(nz=8, dlat_deg=d, dlon_deg=d))
    L = BandLayout(GR.ny, nb_grid)
    F = ModelFields(GR)
    initialize_fields(GR, F)
    core = HipDycore(GR.nx, L.ny(band), GR.nz, pair_top=GR.nl.pair_top, device=device, ny_global=GR.ny,
                     j_offset=L.j_offset(band), dtype=dtype)
    core.set_grid(band_grid_arrays(GR.GRF['HIP'], L, band))
    for n in ('HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB'):
        core.upload(n, L.slice_rows(F.host[n], band, FIELD_TABLE[n][1]))
    core.primary_diag()
    nb = NativeBands.rccl(core, unique_id, rank, nranks, None if south < 0 else south, None if north < 0 else north)
    nb.connect()
    nb.step_matsuno(GR.dt, 2)
    core.sync()
    u = core.download('UWIND')
    if not np.isfinite(u[1:GR.nx + 1, 1:-1]).all():
        raise SystemExit('non-finite winds after two band steps')
    nb.close()
    core.close()
    print('BANDS_PROBE_OK', flush=True)


if __name__ == '__main__':
    main(sys.argv[1:])
