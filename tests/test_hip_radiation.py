"""Radiation columns on the GPU: dycore_radiation through the reference's
Radiation interface against the reference's own output (golden rad_* vectors) and the CPU oracle."""
import datetime

import numpy as np
import pytest

from golden_util import load_golden
from oracle_binding import oracle_radiation, RAD_INPUTS, RAD_OUTPUTS
from hip_util import grid_from_golden

from climate_model_amd import Namelist, Grid, ModelFields, HipDycore, Radiation, DycoreError
from climate_model_amd.synthetic import initialize_fields, initialize_moisture

pytestmark = pytest.mark.gpu
DEVICE_INPUTS = ['PHIVB', 'TAIR', 'RHO', 'SOILTEMP', 'SURFALBEDLW', 'SURFALBEDSW', 'QC']
# max-norm, relative to the largest flux of the scheme (the diffuse shortwave flux is a difference of
# solution terms ten times its size): device exp/pow and FMA contraction through two column solves
TOL = 1e-10
# flux divergences and the heating rate difference ~1000 W/m2 fluxes over layers that absorb 0.1-1 % of
# them: the same absolute flux error is 100-1000 times larger relative to the divergence
TOL_DIV = 1e-8
DIVS = ('SWFLXDIV', 'LWFLXDIV', 'TOTFLXDIV', 'dPOTTdt_RAD')


@pytest.mark.parametrize('case', ['g10x4_rad', 'g4x8_rad'])
def test_radiation_vs_reference(case):
    z = load_golden(case)
    GR = grid_from_golden(z)
    GR.lon_rad = z['rad_meta/lon_rad']
    core = HipDycore(GR.nx, GR.ny, GR.nz, i_radiation=1)
    core.set_grid(GR.GRF['HIP'])
    F = ModelFields(GR, core)
    for n in DEVICE_INPUTS:
        F.host[n][...] = z['rad_in/' + n][:, :, :F.host[n].shape[2]]
        core.upload(n, F.host[n])
    RAD = Radiation(GR)
    np.testing.assert_allclose(RAD.planck_lambdas_center, z['rad_meta/planck_lambdas_center'], rtol=1e-15)
    np.testing.assert_allclose(RAD.planck_dlambdas, z['rad_meta/planck_dlambdas'], rtol=1e-14)
    gmt = datetime.datetime(*[int(v) for v in z['rad_meta/gmt']])
    RAD.simple_radiation(GR, host=F.host, current_GMT=gmt, **F.get(RAD.fields_radiation, target='HIP'))
    assert abs(RAD.solar_constant - float(z['rad_meta/solar_constant'])) < 1e-12 * RAD.solar_constant
    for n in ('SOLZEN', 'MYSUN', 'SWINTOA'):
        ref = z['rad_in/' + n][1:-1, 1:-1]
        assert np.max(np.abs(F.host[n][1:-1, 1:-1] - ref)) <= 1e-13 * np.max(np.abs(ref)), n
    fscale = max(np.max(np.abs(z['rad_out/' + n][1:-1, 1:-1])) for n in RAD_OUTPUTS if n not in DIVS)
    for n in RAD_OUTPUTS:
        got, ref = core.download(n)[1:-1, 1:-1], z['rad_out/' + n][1:-1, 1:-1]
        assert np.isfinite(got).all(), n
        scale = np.max(np.abs(ref)) if n in DIVS else fscale
        assert np.max(np.abs(got - ref)) <= (TOL_DIV if n in DIVS else TOL) * scale, (n, np.max(np.abs(got - ref)) / scale)
    core.close()


def test_radiation_needs_the_switch():
    core = HipDycore(36, 18, 4)
    with pytest.raises(DycoreError, match='i_radiation'):
        core.upload('SOILTEMP', np.zeros((38, 20, 1)))
    with pytest.raises(DycoreError, match='i_radiation'):
        core.radiation(1365., [1e-5], [-1e-6])
    core.close()


@pytest.mark.parametrize('d,nz', [(2.0, 16), (1.0, 32), (0.5, 64)])
def test_radiation_vs_oracle_larger_grids(d, nz):
    """Synthetic state, every column: device against the CPU restatement; several workspace chunks
    at the largest size."""
    GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d, i_radiation=1))
    core = HipDycore(GR.nx, GR.ny, GR.nz, i_radiation=1, i_moist=1)
    core.set_grid(GR.GRF['HIP'])
    F = ModelFields(GR, core)
    initialize_fields(GR, F)
    initialize_moisture(GR, F)
    lat = GR.lat_rad[:, :, 0]
    F.host['QC'][...] *= 400.                       # up to ~0.008: beyond the 0.002 cap
    F.host['SOILTEMP'][:, :, 0] = 295 - np.sin(lat) ** 2 * 30 - F.host['HSURF'][:, :, 0] * 0.01     # srfc_main.py:57-58
    F.host['SURFALBEDLW'][...] = 0.
    F.host['SURFALBEDSW'][...] = np.where(F.host['SOILTEMP'] <= 273.15, 0.6, 0.2)
    F.copy_host_to_device()
    core.primary_diag()
    core.secondary_diag()
    RAD = Radiation(GR)
    gmt = datetime.datetime(2019, 6, 21, 14, 0, 0)
    RAD.simple_radiation(GR, host=F.host, current_GMT=gmt, **F.get(RAD.fields_radiation, target='HIP'))
    inp = {n: core.download(n) for n in RAD_INPUTS}
    ref = oracle_radiation(GR.nx, GR.ny, GR.nz, inp, RAD.solar_constant, RAD.planck_lambdas_center, RAD.planck_dlambdas)
    sun = inp['MYSUN'][1:-1, 1:-1, 0] > 0
    assert 0.3 < sun.mean() < 0.7
    fscale = max(np.max(np.abs(ref[n][1:-1, 1:-1])) for n in RAD_OUTPUTS if n not in DIVS)
    for n in RAD_OUTPUTS:
        got, r = core.download(n)[1:-1, 1:-1], ref[n][1:-1, 1:-1]
        assert np.isfinite(got).all(), n
        scale = np.max(np.abs(r)) if n in DIVS else fscale
        assert np.max(np.abs(got - r)) <= (TOL_DIV if n in DIVS else TOL) * scale, (n, np.max(np.abs(got - r)) / scale)
    heat = core.download('dPOTTdt_RAD')[1:-1, 1:-1]
    assert np.abs(heat).max() < 1e-2               # K/s: a sane heating rate
    core.close()


def test_solver_loop_with_radiation_and_moisture_vs_oracle():
    """BASELINE config 5 without the surface scheme: moist dycore + radiation heating through
    solver.run (radiation at the steps rad_nth_hour selects, then the fused Matsuno step with
    i_POTT_radiation), against the oracle driven the same way."""
    from climate_model_amd import solver
    from oracle_binding import Oracle
    nl = Namelist(nz=8, dlat_deg=4, dlon_deg=4, i_radiation=1, i_moist_main_switch=1, rad_nth_hour=0.02)
    GR = Grid(nl)
    assert nl.i_POTT_radiation == 1
    core = HipDycore(GR.nx, GR.ny, GR.nz, i_radiation=1, i_moist=1)
    core.set_grid(GR.GRF['HIP'])
    F = ModelFields(GR, core)
    initialize_fields(GR, F)
    initialize_moisture(GR, F)
    lat = GR.lat_rad[:, :, 0]
    F.host['SOILTEMP'][:, :, 0] = 295 - np.sin(lat) ** 2 * 30
    F.host['SURFALBEDLW'][...] = 0.
    F.host['SURFALBEDSW'][...] = 0.2
    F.host['dPOTTdt_RAD'][...] = 0.
    F.copy_host_to_device()
    state = {n: F.host[n].copy() for n in ('HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB', 'QV', 'QC', 'SLHFLX')}
    gmt0 = GR.GMT
    nsteps = 4
    RAD = Radiation(GR, nl.rad_nth_hour)
    rad_steps = [s for s in range(1, nsteps + 1) if s % RAD.rad_nth_ts == 0]
    assert 0 < len(rad_steps) < nsteps, (RAD.rad_nth_ts, GR.dt)       # some steps with, some without a radiation call
    solver.run(GR, F, nsteps, nth_ts_print_diag=1000)
    assert GR.ts == nsteps and GR.GMT == gmt0 + datetime.timedelta(seconds=GR.dt * nsteps)

    o = Oracle(GR.nx, GR.ny, GR.nz, GR.GRF['CPU'], state, pair_top=nl.pair_top, i_turb=0, i_radiation=1, i_moist=1)
    o.F['dPOTTdt_RAD'][...] = 0.
    o.primary_diag()
    GR2 = Grid(nl)
    for s in range(0, nsteps + 1):           # s = 0: the radiation call on the initial state (main_fields.py:94-101)
        o.secondary_diag()
        if s in rad_steps or s == 0:
            gmt = gmt0 + datetime.timedelta(seconds=GR.dt * s)
            solzen = rad_solar_zenith_angle_host(GR2, gmt)
            inp = {n: o.F[n] for n in ('PHIVB', 'TAIR', 'RHO', 'QC')}
            inp.update(SOILTEMP=F.host['SOILTEMP'], SURFALBEDLW=F.host['SURFALBEDLW'], SURFALBEDSW=F.host['SURFALBEDSW'],
                       MYSUN=np.cos(solzen), SWINTOA=solar_constant_host(gmt) * np.cos(solzen))
            out = oracle_radiation(GR.nx, GR.ny, GR.nz, inp, solar_constant_host(gmt), RAD.planck_lambdas_center, RAD.planck_dlambdas)
            o.F['dPOTTdt_RAD'][1:-1, 1:-1] = out['dPOTTdt_RAD'][1:-1, 1:-1]
        if s > 0:
            o.step_matsuno(GR.dt, 1, with_rho=False)
    from golden_util import rel_max_err
    for n in ('COLP', 'UWIND', 'VWIND', 'POTT', 'QV'):
        err, ncmp, bad = rel_max_err(core.download(n), o.F[n], n, GR.nx, GR.ny)
        assert bad == 0 and ncmp > 0 and err < 1e-10, (n, err)
    # the heating did act: same run without radiation ends elsewhere
    heat = core.download('dPOTTdt_RAD')[1:-1, 1:-1]
    assert np.isfinite(heat).all() and np.abs(heat).max() > 1e-7
    core.close()


def rad_solar_zenith_angle_host(GR, gmt):
    from climate_model_amd.radiation import rad_solar_zenith_angle
    return rad_solar_zenith_angle(GR, gmt, np.full((GR.nx + 2, GR.ny + 2, 1), np.nan))


def solar_constant_host(gmt):
    from climate_model_amd.radiation import calc_current_solar_constant
    return calc_current_solar_constant(gmt)


@pytest.mark.parametrize('nbands', [2, 5])
def test_band_radiation_equals_single(nbands):
    """Latitude bands: moist step + radiation heating per band = the single-domain run, bit for bit."""
    import torch
    from climate_model_amd.bands import BandLayout, LocalBands, band_grid_arrays, gather_global
    from climate_model_amd.radiation import band_radiation
    from climate_model_amd.fields import FIELD_TABLE
    GR = Grid(Namelist(nz=8, dlat_deg=4, dlon_deg=4, i_radiation=1, i_moist_main_switch=1))
    F = ModelFields(GR)
    initialize_fields(GR, F)
    initialize_moisture(GR, F)
    F.host['SOILTEMP'][:, :, 0] = 295 - np.sin(GR.lat_rad[:, :, 0]) ** 2 * 30
    F.host['SURFALBEDLW'][...] = 0.
    F.host['SURFALBEDSW'][...] = 0.2
    names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB', 'QV', 'QC', 'SLHFLX',
             'SOILTEMP', 'SURFALBEDLW', 'SURFALBEDSW']
    gmt = datetime.datetime(2019, 3, 1, 6, 0, 0)
    RAD = Radiation(GR)
    check = ['COLP', 'UWIND', 'VWIND', 'POTT', 'QV', 'dPOTTdt_RAD', 'LWFLXNET', 'SWFLXNET']

    def prepare(core, grf, sl):
        core.set_grid(grf)
        for n in names:
            core.upload(n, sl(F.host[n], n))
        core.primary_diag()

    single = HipDycore(GR.nx, GR.ny, GR.nz, i_radiation=1, i_moist=1)
    prepare(single, GR.GRF['HIP'], lambda a, n: a)
    single.secondary_diag()
    L1 = BandLayout(GR.ny, 1)
    band_radiation(RAD, GR, L1, 0, single, gmt)
    single.step_matsuno(GR.dt, 2)
    ref = {n: single.download(n) for n in check}
    single.close()

    L = BandLayout(GR.ny, nbands)
    stream = torch.cuda.Stream()
    cores = []
    for r in range(nbands):
        c = HipDycore(GR.nx, L.ny(r), GR.nz, ny_global=GR.ny, j_offset=L.j_offset(r), i_radiation=1, i_moist=1,
                      stream=stream.cuda_stream)
        prepare(c, band_grid_arrays(GR.GRF['HIP'], L, r), lambda a, n, r=r: L.slice_rows(a, r, FIELD_TABLE[n][1]))
        c.secondary_diag()
        band_radiation(RAD, GR, L, r, c, gmt)
        cores.append(c)
    LocalBands(cores, L, torch.device('cuda:0')).step_matsuno(GR.dt, 2)
    for n in check:
        got = gather_global(L, [c.download(n) for c in cores], n)
        np.testing.assert_array_equal(got[1:-1, 1:-1], ref[n][1:-1, 1:-1], err_msg='%s (%d bands)' % (n, nbands))
    for c in cores:
        c.close()
