"""Helpers for the -m gpu parity tests: build a HipDycore + ModelFields from a golden
fixture or from an Oracle state."""
import numpy as np

from climate_model_amd import Grid, ModelFields, HipDycore
from climate_model_amd.fields import FIELD_TABLE
from oracle_binding import GRID_2D, GRID_1D


def grid_from_golden(z, i_turb=0, i_moist=0):
    nx, ny, nz = int(z['meta/nx']), int(z['meta/ny']), int(z['meta/nz'])
    grf = {n: z['grid/' + n] for n in GRID_2D + GRID_1D}
    return Grid.from_arrays(nx, ny, nz, grf, dt=float(z['meta/dt']),
                            pair_top=float(z['meta/pair_top']), i_turb=i_turb, i_moist=i_moist)


def make_core(GR, i_turb=0, i_moist=0, dtype=np.float64, term_off=0):
    core = HipDycore(GR.nx, GR.ny, GR.nz, pair_top=GR.nl.pair_top, i_turb=i_turb, i_moist=i_moist, dtype=dtype, term_off=term_off)
    core.set_grid(GR.GRF['HIP'])
    return core


def fields_from(GR, core, values):
    """ModelFields whose host arrays are overlaid with `values` (dict name->array) and
    uploaded."""
    F = ModelFields(GR, core)
    for n, a in values.items():
        if n in FIELD_TABLE:
            F.host[n][...] = a
    F.copy_host_to_device()
    return F
