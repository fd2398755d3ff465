"""History output (NetCDF-3 files in the reference's layout), restart archives and the
testsuite.py comparison: host-side logic, no GPU (a stand-in core serves the downloads)."""
import os

import numpy as np
import pytest
from scipy.io import netcdf_file

from climate_model_amd import Namelist, Grid, ModelFields
from climate_model_amd import io_output
from climate_model_amd.fields import FIELD_TABLE


class HostCore:
    """Stands in for HipDycore: the 'device' copy is the host array itself."""
    i_moist = True

    def __init__(self):
        self.uploaded, self.diag_calls = [], 0

    def download(self, name, out=None):
        return out

    def upload(self, name, a):
        self.uploaded.append(name)

    def secondary_diag(self):
        self.diag_calls += 1

    def primary_diag(self):
        self.diag_calls += 1


def _filled(nl):
    GR = Grid(nl)
    core = HostCore()
    F = ModelFields(GR)
    F.core = core
    rng = np.random.default_rng(5)
    for n in FIELD_TABLE:
        F.host[n][...] = rng.uniform(0.5, 2.0, F.host[n].shape)
    F.host['PHIVB'][...] = np.cumsum(rng.uniform(100., 3000., F.host['PHIVB'].shape)[:, :, ::-1], axis=2)[:, :, ::-1]
    return GR, F, core


def test_history_file_layout_and_values(tmp_path):
    GR, F, core = _filled(Namelist(nz=5, dlat_deg=10, dlon_deg=10, i_radiation=1))
    GR.ts = 7
    fn = io_output.output_to_NC(GR, F, str(tmp_path))
    assert os.path.basename(fn) == 'out0001.nc' and GR.nc_output_count == 1 and core.diag_calls == 1
    f = netcdf_file(fn, 'r', mmap=False)
    d = f.dimensions
    assert (d['lon'], d['lons'], d['lat'], d['lats'], d['level'], d['levels']) == (36, 37, 18, 19, 5, 6)
    assert abs(f.variables['time'].data[0] - 7 * GR.dt / 86400) < 1e-12
    # orientation (time, level, lat, lon), interior cells, float32 (io_nc_output.py:86-117)
    u = f.variables['UWIND']
    assert u.dimensions == ('time', 'level', 'lat', 'lons') and u.data.dtype.itemsize == 4
    np.testing.assert_array_equal(u.data[0], F.host['UWIND'][1:-1, 1:-1, :].T.astype(np.float32))
    v = f.variables['VWIND']
    assert v.dimensions == ('time', 'level', 'lats', 'lon')
    np.testing.assert_array_equal(f.variables['COLP'].data[0], F.host['COLP'][1:-1, 1:-1, 0].T.astype(np.float32))
    np.testing.assert_allclose(f.variables['PSURF'].data[0], F.host['COLP'][1:-1, 1:-1, 0].T + 10000., rtol=1e-6)
    np.testing.assert_allclose(f.variables['WWIND'].data[0],
                               (F.host['WWIND'][1:-1, 1:-1, :] * F.host['COLP'][1:-1, 1:-1, :]).T, rtol=1e-6)
    # zonal-mean profile, bottom-up (io_nc_output.py:203-207)
    np.testing.assert_allclose(f.variables['UWINDprof'].data[0, 0], np.mean(F.host['UWIND'][1:-1, 1:-1, 4], axis=0), rtol=1e-6)
    # radiation block and the K/h conversion of the heating rate (io_nc_output.py:293-308)
    np.testing.assert_allclose(f.variables['dPOTTdt_RAD'].data[0], F.host['dPOTTdt_RAD'][1:-1, 1:-1, :].T * 3600, rtol=1e-6)
    assert f.variables['LWFLXNET'].dimensions == ('time', 'levels', 'lat', 'lon')
    # vorticity of io_functions.py:22-37 at one cell
    i, j, k = 4, 6, 2
    U, V = F.host['UWIND'][:, :, k], F.host['VWIND'][:, :, k]
    vort = (((V[i + 1, j] + V[i + 1, j + 1]) / 2 - (V[i - 1, j] + V[i - 1, j + 1]) / 2) / (2 * GR.dx[i, j, 0])
            - ((U[i, j + 1] + U[i + 1, j + 1]) / 2 - (U[i, j - 1] + U[i + 1, j - 1]) / 2) / (2 * GR.dlat_rad_1D * 6371000.))
    assert abs(f.variables['VORT'].data[0, k, j - 1, i - 1] - vort) < 1e-6 * abs(vort)
    # water-vapour path: sum of QV * dz * RHO (io_functions.py:55-58)
    dz = (F.host['PHIVB'][i, j, :-1] - F.host['PHIVB'][i, j, 1:]) / 9.81
    assert abs(f.variables['WVP'].data[0, j - 1, i - 1] - np.sum(F.host['QV'][i, j] * dz * F.host['RHO'][i, j])) < 1e-5 * f.variables['WVP'].data[0, j - 1, i - 1]
    assert 'POTT' not in f.variables            # switched off in the reference's output_fields
    f.close()
    cf = io_output.constant_fields_to_NC(GR, F, str(tmp_path))
    f = netcdf_file(cf, 'r', mmap=False)
    np.testing.assert_array_equal(f.variables['HSURF'].data, F.host['HSURF'][1:-1, 1:-1, 0].T.astype(np.float32))
    f.close()


def test_testsuite_criterion(tmp_path):
    GR, F, _ = _filled(Namelist(nz=4, dlat_deg=10, dlon_deg=10))
    a = io_output.output_to_NC(GR, F, str(tmp_path / 'ref'), count=2)
    failed, devs = io_output.compare_outputs(a, a, out=lambda s: None)
    assert not failed and all(v == 0 for v in devs.values()) and {'UWIND', 'VWIND', 'WWIND', 'COLP', 'PHI', 'QV', 'QC'} <= set(devs)
    F.host['UWIND'][5, 5, 1] *= 1 + 3e-4
    b = io_output.output_to_NC(GR, F, str(tmp_path / 'test'), count=2)
    failed, devs = io_output.compare_outputs(a, b, out=lambda s: None)
    assert failed and devs['UWIND'] > 1e-4 and devs['VWIND'] == 0


def test_restart_round_trip(tmp_path):
    import datetime
    nl = Namelist(nz=4, dlat_deg=10, dlon_deg=10)
    GR, F, core = _filled(nl)
    GR.ts, GR.nc_output_count = 123, 4
    GR.GMT = datetime.datetime(2019, 2, 3, 4, 5, 6)
    path = io_output.write_restart(GR, F, str(tmp_path))
    assert os.path.basename(path) == '10_10_4.npz'
    GR2 = Grid(nl)
    F2 = ModelFields(GR2)
    F2.core = HostCore()
    io_output.load_restart(GR2, F2, path)
    assert (GR2.ts, GR2.nc_output_count, GR2.GMT) == (123, 4, GR.GMT)
    for n in io_output.RESTART_FIELDS:
        np.testing.assert_array_equal(F2.host[n], F.host[n])
    assert set(F2.core.uploaded) == set(io_output.RESTART_FIELDS) and F2.core.diag_calls == 1
    GR3 = Grid(Namelist(nz=4, dlat_deg=10, dlon_deg=10), sigma_vb=np.linspace(0, 1, 5))
    with pytest.raises(ValueError, match='sigma'):
        io_output.load_restart(GR3, ModelFields(GR3), path)


def test_time_step_rule_of_the_reference_presets():
    """main_grid.py:258-270: dt = int(CFL min(dx) / 400), lowered until it divides the output interval; the step
    counts derived from it.  The two presets of namelist.py:243-278 (values from running the reference's own Grid
    in the build container: dt = 200 s for preset 1 -- meta/dt of tests/golden/g5x8_preset.npz -- and 20 s for preset 2)."""
    from climate_model_amd import Namelist, Grid
    g = Grid(Namelist(nz=8, dlat_deg=5, dlon_deg=5, lat0_deg=-80, lat1_deg=80, i_out_nth_hour=8, i_sim_n_days=0.72))
    assert (g.dt, g.i_out_nth_ts, g.i_restart_nth_ts) == (200, 144, 432) and abs(g.nts - 311.04) < 1e-9
    g = Grid(Namelist(nz=32, dlat_deg=1, dlon_deg=1, lat0_deg=-84, lat1_deg=84, i_out_nth_hour=12))
    assert (g.dt, g.i_out_nth_ts, g.i_restart_nth_ts) == (20, 2160, 4320)
    assert 12 * 3600 % g.dt == 0
    # an explicit dt is taken as is
    assert Grid(Namelist(nz=8, dlat_deg=5, dlon_deg=5, dt=37.5)).dt == 37.5


def test_restart_refuses_another_time_step(tmp_path):
    from climate_model_amd import Namelist, Grid, ModelFields, io_output
    GR = Grid(Namelist(nz=4, dlat_deg=10, dlon_deg=10))

    class _Core:
        i_moist = False

        def download(self, n, out):
            return out
    F = ModelFields(GR)
    F.core = _Core()
    path = io_output.write_restart(GR, F, str(tmp_path))
    F.core = None
    GR2 = Grid(Namelist(nz=4, dlat_deg=10, dlon_deg=10, dt=GR.dt / 2))
    with pytest.raises(ValueError, match='dt'):
        io_output.load_restart(GR2, ModelFields(GR2), path)
    io_output.load_restart(GR, F, path)
