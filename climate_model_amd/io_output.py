"""History output, restart files and the output comparison of the reference.

Counterpart of `io_nc_output.py` (`output_to_NC`, `constant_fields_to_NC`), the output
diagnostics of `io_functions.NC_output_diagnostics` (:18-61), `io_restart.py` and `testsuite.py`.
Files are NetCDF-3 (`scipy.io.netcdf_file`; the reference writes NetCDF-4 through netCDF4, which
this environment does not have) with the reference's file names, dimensions, variable names,
orientation (time, level, lat, lon) and float32 payload, so the reference's own `testsuite.py`
reads them.  Restart files are `.npz` archives of the prognostic state instead of pickles of the
whole Grid/ModelFields objects (nothing executable in the file).

Host-side I/O: fields are downloaded from the device first.
"""
import copy
import os

import numpy as np
from scipy.io import netcdf_file

from .constants import con_g, con_rE
from .fields import FIELD_TABLE

# namelist.py:150-216 restricted to the fields this package owns; flags > 1 add the zonal-mean profile
OUTPUT_FIELDS = {
    'PSURF': 1, 'COLP': 1, 'UWIND': 2, 'VWIND': 2, 'WIND': 2, 'WWIND': 2, 'VORT': 1, 'POTT': 0, 'TAIR': 2,
    'PHI': 1, 'PAIR': 0, 'RHO': 0, 'SSHFLX': 1, 'SLHFLX': 1, 'SMOMXFLX': 0, 'SMOMYFLX': 0, 'QV': 2, 'QC': 1,
    'dQVdt': 1, 'dQVdt_TURB': 1, 'dUFLXdt_TURB': 1, 'dVFLXdt_TURB': 1, 'dPOTTdt_TURB': 1, 'KMOM': 1, 'KHEAT': 1,
    'KMOM_dUWINDdz': 0, 'KMOM_dVWINDdz': 0, 'WVP': 1, 'CWP': 1,
}
NC_OUT_DIAG_FIELDS = ['UWIND', 'VWIND', 'WWIND', 'POTT', 'COLP', 'PVTF', 'PVTFVB', 'PHI', 'PHIVB', 'RHO', 'QV', 'QC']
RADIATION_OUT = ['SWDIFFLXDO', 'SWDIRFLXDO', 'SWFLXUP', 'SWFLXDO', 'SWFLXNET', 'LWFLXUP', 'LWFLXDO', 'LWFLXNET']
RESTART_FIELDS = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB', 'QV', 'QC']


def NC_output_diagnostics(GR, UWIND, VWIND, WWIND, POTT, COLP, PVTF, PVTFVB, PHI, PHIVB, RHO, QV, QC):
    """io_functions.py:18-61: relative vorticity, vertical wind in m/s, water-vapour and cloud-water path."""
    ii, jj = GR.ii, GR.jj
    VORT = np.full((GR.nx + 2, GR.ny + 2, GR.nz), np.nan)
    dy = GR.dlat_rad_1D * con_rE                      # main_grid.py:219
    for k in range(GR.nz):
        V, U = VWIND[:, :, k], UWIND[:, :, k]
        VORT[:, :, k][ii, jj] = ((+ (V[ii + 1, jj] + V[ii + 1, jj + 1]) / 2 - (V[ii - 1, jj] + V[ii - 1, jj + 1]) / 2)
                                 / (2 * GR.dx[ii, jj, 0])
                                 - (+ (U[ii, jj + 1] + U[ii + 1, jj + 1]) / 2 - (U[ii, jj - 1] + U[ii + 1, jj - 1]) / 2)
                                 / (2 * dy))
    WWIND_ms = copy.deepcopy(WWIND)
    for ks in range(1, GR.nzs - 1):
        WWIND_ms[:, :, ks][ii, jj] = ((PHI[:, :, ks][ii, jj] - PHI[:, :, ks - 1][ii, jj]) /
                                      (con_g * 0.5 * (GR.dsigma[0, 0, ks] + GR.dsigma[0, 0, ks - 1])) *
                                      WWIND[:, :, ks][ii, jj])
    ALTVB = PHIVB / con_g
    dz = ALTVB[:, :, :-1][ii, jj] - ALTVB[:, :, 1:][ii, jj]
    WVP = np.sum(QV[ii, jj] * dz * RHO[ii, jj], 2)
    CWP = np.sum(QC[ii, jj] * dz * RHO[ii, jj], 2)
    return VORT, WWIND_ms, WVP, CWP


def _dims(ncf, GR, with_time):
    if with_time:
        ncf.createDimension('time', None)
    for name, n in (('lon', GR.nx), ('lons', GR.nxs), ('lat', GR.ny), ('lats', GR.nys), ('level', GR.nz), ('levels', GR.nzs)):
        ncf.createDimension(name, n)
    nb = GR.nb
    coords = (('lon', GR.lon_rad[GR.ii, nb + 1, 0]), ('lons', GR.lon_is_rad[GR.iis, nb + 1, 0]),
              ('lat', GR.lat_rad[nb + 1, GR.jj, 0]), ('lats', GR.lat_js_rad[nb + 1, GR.jjs, 0]),
              ('level', np.arange(GR.nz)), ('levels', np.arange(GR.nzs)))
    for name, val in coords:
        v = ncf.createVariable(name, 'f4', (name,))
        v[:] = np.asarray(val, dtype=np.float32).reshape(-1)


def _put(ncf, name, dims, data):
    v = ncf.createVariable(name, 'f4', ('time',) + dims)
    v[0] = np.asarray(data, dtype=np.float32)


def output_to_NC(GR, F, output_path, count=None, output_fields=None, i_radiation=None, refresh_diag=True):
    """io_nc_output.py:30-317.  Downloads what it writes; returns the file name.  `refresh_diag`: re-diagnose TAIR,
    PAIR, RHO, WIND from the current state first; the reference writes what the loop's last secondary_diag call left
    (the state before the last step's dynamics, solver.py:98-101 vs :152-164), which is what solver.run asks for."""
    of = dict(OUTPUT_FIELDS if output_fields is None else output_fields)
    i_radiation = GR.nl.i_radiation if i_radiation is None else i_radiation
    count = getattr(GR, 'nc_output_count', 0) + 1 if count is None else count      # solver.py:162
    core = F.core
    moist = bool(getattr(core, 'i_moist', False))
    want = set(NC_OUT_DIAG_FIELDS) | {n for n, flag in of.items() if flag and n in FIELD_TABLE} | {'WIND', 'TAIR'}
    if i_radiation:
        want |= set(RADIATION_OUT) | {'dPOTTdt_RAD'}
    if not moist:
        want -= {'QV', 'QC', 'dQVdt', 'dQVdt_TURB'}
    if refresh_diag:
        core.secondary_diag()
    for n in sorted(want):
        core.download(n, F.host[n])
    h = F.host
    zeros = None
    if not moist:
        zeros = np.zeros((GR.nx + 2, GR.ny + 2, GR.nz))
    diag_in = {n: (h[n] if (moist or n not in ('QV', 'QC')) else zeros) for n in NC_OUT_DIAG_FIELDS}
    VORT, WWIND_ms, WVP, CWP = NC_output_diagnostics(GR, **diag_in)

    os.makedirs(output_path, exist_ok=True)
    filename = os.path.join(output_path, 'out' + str(count).zfill(4) + '.nc')
    ncf = netcdf_file(filename, 'w', version=2)
    _dims(ncf, GR, True)
    t = ncf.createVariable('time', 'f8', ('time',))
    t[0] = GR.ts * GR.dt / 3600 / 24
    ii, jj = GR.ii, GR.jj
    for name in ('UWIND', 'VWIND', 'WIND', 'POTT', 'TAIR', 'PHI', 'PAIR', 'RHO', 'COLP', 'QV', 'QC', 'dQVdt', 'dQVdt_TURB',
                 'dUFLXdt_TURB', 'dVFLXdt_TURB', 'dPOTTdt_TURB', 'KMOM', 'KHEAT', 'SMOMXFLX', 'SMOMYFLX', 'SSHFLX',
                 'SLHFLX', 'KMOM_dUWINDdz', 'KMOM_dVWINDdz'):
        if not of.get(name) or name not in want:
            continue
        a = h[name]
        dimx, dimy, dimz = a.shape
        lon_str = 'lon' if dimx == GR.nx + 2 else 'lons'
        lat_str = 'lat' if dimy == GR.ny + 2 else 'lats'
        i = np.arange(1, dimx - 1)
        j = np.arange(1, dimy - 1)
        sel = a[np.ix_(i, j)]
        if dimz == 1:
            _put(ncf, name, (lat_str, lon_str), sel[:, :, 0].T)
        else:
            _put(ncf, name, ('level' if dimz == GR.nz else 'levels', lat_str, lon_str), sel.T)
    if of.get('PSURF'):
        _put(ncf, 'PSURF', ('lat', 'lon'), h['COLP'][ii, jj, 0].T + GR.nl.pair_top)
    if of.get('WWIND'):
        _put(ncf, 'WWIND', ('levels', 'lat', 'lon'), (h['WWIND'][ii, jj, :] * h['COLP'][ii, jj, :]).T)
    if of.get('VORT'):
        _put(ncf, 'VORT', ('level', 'lat', 'lon'), VORT[ii, jj].T)
    if of.get('WVP') and moist:
        _put(ncf, 'WVP', ('lat', 'lon'), WVP.T)
    if of.get('CWP') and moist:
        _put(ncf, 'CWP', ('lat', 'lon'), CWP.T)
    # zonal-mean profiles, stored bottom-up as the reference does (io_nc_output.py:175-232)
    prof = (('UWIND', 'level', 'lat', (GR.iis, GR.jj)), ('VWIND', 'level', 'lats', (GR.ii, GR.jjs)),
            ('POTT', 'level', 'lat', (ii, jj)), ('TAIR', 'level', 'lat', (ii, jj)), ('QV', 'level', 'lat', (ii, jj)),
            ('QC', 'level', 'lat', (ii, jj)), ('WWIND', 'levels', 'lat', (ii, jj)))
    for name, lev, lat_str, sel in prof:
        if of.get(name, 0) > 1 and name in want:
            _put(ncf, name + 'prof', (lev, lat_str), np.mean(h[name][sel], axis=0)[:, ::-1].T)
    if of.get('VORT', 0) > 1:
        _put(ncf, 'VORTprof', ('level', 'lat'), np.mean(VORT[ii, jj], axis=0)[:, ::-1].T)
    if i_radiation:
        for name in RADIATION_OUT:
            _put(ncf, name, ('levels', 'lat', 'lon'), h[name][ii, jj, :].T)
        _put(ncf, 'dPOTTdt_RAD', ('level', 'lat', 'lon'), h['dPOTTdt_RAD'][ii, jj, :].T * 3600)
    ncf.close()
    GR.nc_output_count = count
    return filename


def constant_fields_to_NC(GR, F, output_path):
    """io_nc_output.py:320-367"""
    os.makedirs(output_path, exist_ok=True)
    filename = os.path.join(output_path, 'constants.nc')
    ncf = netcdf_file(filename, 'w', version=2)
    _dims(ncf, GR, False)
    v = ncf.createVariable('HSURF', 'f4', ('lat', 'lon'))
    v[:] = np.asarray(F.host['HSURF'][GR.ii, GR.jj, 0].T, dtype=np.float32)
    ncf.close()
    return filename


def write_restart(GR, F, directory):
    """io_restart.py:24-55, as an .npz archive of the state (device copy downloaded first)."""
    os.makedirs(directory, exist_ok=True)
    names = [n for n in RESTART_FIELDS if n not in ('QV', 'QC') or getattr(F.core, 'i_moist', False)]
    for n in names:
        F.core.download(n, F.host[n])
    path = os.path.join(directory, '%s_%s_%d.npz' % (str(GR.dlat_deg).zfill(2), str(GR.dlon_deg).zfill(2), GR.nz))
    np.savez(path, ts=np.int64(GR.ts), dt=np.float64(GR.dt), gmt=np.array(GR.GMT.isoformat()),
             nc_output_count=np.int64(getattr(GR, 'nc_output_count', 0)),
             sigma_vb=GR.sigma_vb.reshape(-1), **{n: F.host[n] for n in names})
    return path


def load_restart(GR, F, path):
    """io_restart.py:58-104: state, step counter and clock back into GR / F (and onto the device)."""
    from datetime import datetime
    z = np.load(path, allow_pickle=False)
    if not np.array_equal(z['sigma_vb'], GR.sigma_vb.reshape(-1)):
        raise ValueError('restart file was written for other sigma levels')
    if 'dt' in z.files and float(z['dt']) != float(GR.dt):
        raise ValueError('restart file was written with dt = %r s, this grid steps with %r s' % (float(z['dt']), float(GR.dt)))
    GR.ts = int(z['ts'])
    GR.GMT = datetime.fromisoformat(str(z['gmt']))
    GR.nc_output_count = int(z['nc_output_count'])
    for n in RESTART_FIELDS:
        if n in z.files:
            if z[n].shape != F.host[n].shape:
                raise ValueError('restart field %s has shape %s, grid needs %s' % (n, z[n].shape, F.host[n].shape))
            F.host[n][...] = z[n]
            if F.core is not None:
                F.core.upload(n, F.host[n])
    if F.core is not None:
        F.core.primary_diag()
    return GR, F


def compare_outputs(ref_file, test_file, test_fields=None, tolerance=1E-4, out=print):
    """testsuite.py:18-80: per-field max deviation relative to the field's maximum; returns
    (failed, {field: deviation})."""
    if test_fields is None:
        test_fields = ['UWIND', 'VWIND', 'WWIND', 'COLP', 'PHI', 'SURFTEMP', 'SLHFLX', 'SMOMXFLX', 'SMOMYFLX',
                       'SURFALBEDSW', 'SSHFLX', 'SWFLXNET', 'LWFLXNET', 'QV', 'QC', 'dVFLXdt_TURB']
    ref, test = netcdf_file(ref_file, 'r', mmap=False), netcdf_file(test_file, 'r', mmap=False)
    failed, devs = False, {}
    for name in test_fields:
        if name not in ref.variables or name not in test.variables:
            out('%s: not in output file.' % name)
            continue
        r, t = np.array(ref.variables[name].data, dtype=np.float64), np.array(test.variables[name].data, dtype=np.float64)
        maxv = np.abs(t).max()
        dev = 0. if maxv <= tolerance else np.abs(t - r).max() / maxv
        devs[name] = dev
        out('%s %.3e%s' % (name, dev, '   !!! Deviation !!!' if dev > tolerance else ''))
        failed = failed or dev > tolerance
    ref.close(); test.close()
    return failed, devs
