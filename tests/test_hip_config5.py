"""BASELINE config 5 (0.25 deg x 64 levels, dyn_moist QV/QC advection + radiative heating) and the moist /
turbulence-coupled kernels at sizes beyond the small golden grids: the HIP path against the CPU oracle.

Tolerances (max-norm relative, testsuite.py:46-60 norm) as in tests/test_hip_parity.py: 1e-10 per run for
COLP, UWIND, VWIND, POTT, QV; 1e-9 for QC (the 1e-7 clamp of the log interpolation, dyn_functions.py:70-95,
amplifies last-bit differences of log())."""
import datetime

import numpy as np
import pytest

from golden_util import rel_max_err, rel_level_err
from oracle_binding import Oracle, oracle_radiation, RAD_INPUTS

from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.bands import BandLayout, LocalBands, band_grid_arrays, gather_global
from climate_model_amd.fields import FIELD_TABLE
from climate_model_amd.radiation import Radiation, band_radiation
from climate_model_amd.synthetic import initialize_fields, initialize_moisture, stable_dt

pytestmark = pytest.mark.gpu
TOL = {'QC': 1e-9}
# turbulence-coupled runs: the scalar fluxes K (q[k-1] - q[k]) / dz / (dz rho) with the random K of these tests
# (0.5 .. 20 m2/s) and the thin upper levels of a 200-level column amplify last-bit differences of the
# quotients: QV measured 1.1e-10 at 40 x 10 x 200, <= 5e-12 everywhere else
TOL_TURB = {'QC': 1e-9, 'QV': 1e-9}
STEP_TOL = 1e-10


def _physics_inputs(GR, F, i_turb, seed=4):
    if i_turb:
        rng = np.random.default_rng(seed)
        for n, lo, hi in [('KMOM', 0.5, 20.), ('KHEAT', 0.5, 20.), ('SSHFLX', -50., 150.),
                          ('SMOMXFLX', -0.2, 0.2), ('SMOMYFLX', -0.2, 0.2)]:
            F.host[n][...] = rng.uniform(lo, hi, F.host[n].shape)


# level by level (golden_util.rel_level_err: scale = max|ref| of that level) on the full-size runs: 1e-9 for every field, observed <= 4.5e-11 (UWIND at the surface level, WWIND), QV / QC 1.6e-15;
# profiles/r03_parity_margins.txt
LEVEL_TOL = {}
LEVEL_STEP_TOL = 1e-9


def _check(core, o, names, nx, ny, what, tol=TOL, per_level=False):
    for n in names:
        got = core.download(n)
        err, ncmp, bad = rel_max_err(got, o.F[n], n, nx, ny)
        assert bad == 0 and ncmp > 0 and err <= tol.get(n, STEP_TOL), '%s %s: rel err %.3e (%d bad)' % (what, n, err, bad)
        if per_level:
            lerr, kw = rel_level_err(got, o.F[n], n, nx, ny)
            assert lerr <= LEVEL_TOL.get(n, LEVEL_STEP_TOL), '%s %s: per-level rel err %.3e at level %d' % (what, n, lerr, kw)


def test_config5_full_size_vs_oracle():
    """0.25 deg x 64 with i_moist = 1 and the radiative heating rate in the temperature tendency, as
    `bench.py --moist --radiation` runs it: device radiation call, then one Matsuno step; the oracle gets
    the same heating rate (the device's, itself checked against the oracle's radiation on a sample of
    columns below) and steps once."""
    nl = Namelist(nz=64, dlat_deg=0.25, dlon_deg=0.25, i_radiation=1, i_moist_main_switch=1)
    GR = Grid(nl)
    F = ModelFields(GR)
    initialize_fields(GR, F)
    initialize_moisture(GR, F)
    lat = GR.lat_rad[:, :, 0]
    F.host['SOILTEMP'][:, :, 0] = 295 - np.sin(lat) ** 2 * 30
    F.host['SURFALBEDLW'][...] = 0.
    F.host['SURFALBEDSW'][...] = 0.2
    names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB', 'QV', 'QC', 'SLHFLX',
             'SOILTEMP', 'SURFALBEDLW', 'SURFALBEDSW']
    core = HipDycore(GR.nx, GR.ny, GR.nz, pair_top=nl.pair_top, i_radiation=1, i_moist=1)
    core.set_grid(GR.GRF['HIP'])
    for n in names:
        core.upload(n, F.host[n])
    core.primary_diag()
    core.secondary_diag()
    RAD = Radiation(GR)
    gmt = datetime.datetime(2019, 3, 21, 12, 0, 0)
    band_radiation(RAD, GR, BandLayout(GR.ny, 1), 0, core, gmt)
    heat = core.download('dPOTTdt_RAD')
    assert np.isfinite(heat[1:-1, 1:-1]).all() and 1e-7 < np.abs(heat[1:-1, 1:-1]).max() < 1e-2

    # radiation at this size: every 97th longitude (15 of 1440) through the oracle's columns
    inp = {n: core.download(n) for n in RAD_INPUTS}
    sel = np.arange(1, GR.nx + 1, 97)
    sub = {n: np.ascontiguousarray(np.concatenate([a[:1], a[sel], a[-1:]], axis=0)) for n, a in inp.items()}
    ref = oracle_radiation(len(sel), GR.ny, GR.nz, sub, RAD.solar_constant, RAD.planck_lambdas_center, RAD.planck_dlambdas)
    r = ref['dPOTTdt_RAD'][1:-1, 1:-1]
    assert np.max(np.abs(heat[sel][:, 1:-1] - r)) <= 1e-9 * np.max(np.abs(r))

    core.step_matsuno(GR.dt, 1)
    state = {n: F.host[n] for n in names if n in ('HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB', 'QV', 'QC', 'SLHFLX')}
    o = Oracle(GR.nx, GR.ny, GR.nz, GR.GRF['CPU'], state, pair_top=nl.pair_top, i_turb=0, i_radiation=1, i_moist=1,
               fill_nan=False)
    o.F['dPOTTdt_RAD'][...] = heat
    o.primary_diag()
    o.step_matsuno(GR.dt, 1, with_rho=False)
    _check(core, o, ['COLP', 'UWIND', 'VWIND', 'POTT', 'QV', 'QC', 'PHI', 'WWIND'], GR.nx, GR.ny, 'config 5', per_level=True)
    core.close()


def _setup_nxny(nx, ny, nz, i_turb, i_moist):
    nl = Namelist(nz=nz, dlat_deg=180.0 / ny * (1 - 1e-12), dlon_deg=360.0 / nx * (1 - 1e-12), dt=1.0,
                  i_UVFLX_vert_turb=i_turb, i_POTT_vert_turb=i_turb, i_moist_main_switch=i_moist)
    GR = Grid(nl)
    assert (GR.nx, GR.ny) == (nx, ny)
    GR.dt = stable_dt(GR)
    F = ModelFields(GR)
    initialize_fields(GR, F)
    _physics_inputs(GR, F, i_turb)
    names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB', 'KMOM', 'KHEAT',
             'SSHFLX', 'SMOMXFLX', 'SMOMYFLX', 'dPOTTdt_RAD']
    if i_moist:
        initialize_moisture(GR, F)
        names += ['QV', 'QC', 'SLHFLX']
    return GR, {n: F.host[n].copy() for n in names}


@pytest.mark.parametrize('i_turb,i_moist', [(1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize('nx,ny,nz', [(65, 9, 5), (67, 16, 8), (130, 17, 16), (200, 33, 7), (96, 25, 64), (70, 12, 96),
                                      (40, 10, 200), (1440, 24, 64), (360, 180, 32)])
def test_ragged_sizes_moist_turbulent_vs_oracle(nx, ny, nz, i_turb, i_moist):
    """The grids of test_ragged_sizes_vs_oracle (nx % 32 != 0, nz = 64, 96, 200, ...) plus a full-width strip
    (1440 columns: the 256-cell linear launch and the XCD order) and BASELINE config 1's grid, with the moisture
    and / or the turbulence-coupled kernels: 2 steps against the oracle."""
    GR, state = _setup_nxny(nx, ny, nz, i_turb, i_moist)
    core = HipDycore(nx, ny, nz, pair_top=GR.nl.pair_top, i_turb=i_turb, i_moist=i_moist)
    core.set_grid(GR.GRF['HIP'])
    for n, a in state.items():
        core.upload(n, a)
    core.primary_diag()
    core.step_matsuno(GR.dt, 2)
    o = Oracle(nx, ny, nz, GR.GRF['CPU'], state, pair_top=GR.nl.pair_top, i_turb=i_turb, i_moist=i_moist)
    o.primary_diag()
    o.step_matsuno(GR.dt, 2, with_rho=bool(i_turb))
    names = ['COLP', 'UWIND', 'VWIND', 'POTT', 'PHI', 'WWIND', 'POTTVB'] + (['QV', 'QC'] if i_moist else [])
    _check(core, o, names, nx, ny, '%dx%dx%d turb=%d moist=%d' % (nx, ny, nz, i_turb, i_moist), TOL_TURB if i_turb else TOL)
    core.close()


def test_turbulent_moist_full_size_vs_oracle():
    """0.25 deg x 64 with the turbulence-coupled and the moisture kernels (random K, surface fluxes): 1 step."""
    nl = Namelist(nz=64, dlat_deg=0.25, dlon_deg=0.25, i_UVFLX_vert_turb=1, i_POTT_vert_turb=1, i_moist_main_switch=1)
    GR = Grid(nl)
    F = ModelFields(GR)
    initialize_fields(GR, F)
    initialize_moisture(GR, F)
    _physics_inputs(GR, F, 1)
    names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB', 'KMOM', 'KHEAT',
             'SSHFLX', 'SMOMXFLX', 'SMOMYFLX', 'dPOTTdt_RAD', 'QV', 'QC', 'SLHFLX']
    state = {n: F.host[n] for n in names}
    core = HipDycore(GR.nx, GR.ny, GR.nz, pair_top=nl.pair_top, i_turb=1, i_moist=1)
    core.set_grid(GR.GRF['HIP'])
    for n, a in state.items():
        core.upload(n, a)
    core.primary_diag()
    core.step_matsuno(GR.dt, 1)
    o = Oracle(GR.nx, GR.ny, GR.nz, GR.GRF['CPU'], state, pair_top=nl.pair_top, i_turb=1, i_moist=1, fill_nan=False)
    o.primary_diag()
    o.step_matsuno(GR.dt, 1, with_rho=True)
    _check(core, o, ['COLP', 'UWIND', 'VWIND', 'POTT', 'QV', 'QC', 'PHI'], GR.nx, GR.ny, 'turbulent moist 0.25 deg', TOL_TURB, per_level=True)
    core.close()


@pytest.mark.parametrize('native', [False, True])
@pytest.mark.parametrize('i_moist', [0, 1])
def test_eight_bands_full_size_bitwise(i_moist, native):
    """0.25 deg x 64 cut into 8 latitude bands of 90 rows (the decomposition of BASELINE configs 4 / 5): two
    steps on the overlapped schedule = the single-domain run, bit for bit."""
    import torch
    nl = Namelist(nz=64, dlat_deg=0.25, dlon_deg=0.25, i_moist_main_switch=i_moist)
    GR = Grid(nl)
    F = ModelFields(GR)
    initialize_fields(GR, F)
    names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB']
    if i_moist:
        initialize_moisture(GR, F)
        names += ['QV', 'QC', 'SLHFLX']
    check = ['COLP', 'UWIND', 'VWIND', 'POTT'] + (['QV', 'QC'] if i_moist else [])
    single = HipDycore(GR.nx, GR.ny, GR.nz, pair_top=nl.pair_top, i_moist=i_moist)
    single.set_grid(GR.GRF['HIP'])
    for n in names:
        single.upload(n, F.host[n])
    single.primary_diag()
    single.step_matsuno(GR.dt, 2)
    ref = {n: single.download(n) for n in check}
    single.close()
    nbands = 8
    L = BandLayout(GR.ny, nbands)
    stream = torch.cuda.Stream()
    cores = []
    for r in range(nbands):
        c = HipDycore(GR.nx, L.ny(r), GR.nz, ny_global=GR.ny, j_offset=L.j_offset(r), i_moist=i_moist,
                      stream=stream.cuda_stream)
        c.set_grid(band_grid_arrays(GR.GRF['HIP'], L, r))
        for n in names:
            c.upload(n, L.slice_rows(F.host[n], r, FIELD_TABLE[n][1]))
        c.primary_diag()
        cores.append(c)
    lb = LocalBands(cores, L, torch.device('cuda:0'), overlap=True, native=native)
    assert lb.overlap and (lb.native is not None) == native
    lb.step_matsuno(GR.dt, 2)
    for n in check:
        got = gather_global(L, [c.download(n) for c in cores], n)
        np.testing.assert_array_equal(got, ref[n], err_msg='%s (8 bands, moist=%d)' % (n, i_moist))
    for c in cores:
        c.close()
