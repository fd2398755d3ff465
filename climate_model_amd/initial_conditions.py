"""Initial conditions and sigma levels the way the reference builds them.

Counterpart of `io_initial_conditions.py`: topography from the 1-degree elevation file
(`load_topo`, :185-213: bilinear interpolation to the mass points, negative heights clipped,
n_topo_smooth passes of a latitude-dependent 5-point smoother), hydrostatic standard-atmosphere
profile (`set_up_profile`, :146-180), sigma levels from equally-spaced-in-z**2 interfaces
(`set_up_sigma_levels`, :253-287), Gaussian wind perturbation (`gaussian2D`, :225-249 and
:62-78).  Needs the reference's two input data files (topography NetCDF-3 file and the
standard-atmosphere table); `data_dir` names the directory that holds them.

The dycore does not need any of this (its inputs are arrays); it exists so that a run can
start from exactly the reference's state.
"""
import os

import numpy as np
from scipy.io import netcdf_file
from scipy.interpolate import RegularGridInterpolator, interp1d

from .constants import con_kappa
from .grid import exchange_BC_host

TOPO_FILE = 'elev.1-deg.nc'
PROFILE_FILE = 'mean_vert_prof.dat'


def _bilinear(x, y, z, xn, yn):
    """The reference's scipy.interpolate.interp2d(x, y, z)(xn, yn) for kind='linear' on a
    regular grid: bilinear inside, nearest outside; returns shape (len(yn), len(xn)) with xn, yn
    sorted ascending."""
    x = np.asarray(x, dtype=np.float64).ravel()
    y = np.asarray(y, dtype=np.float64).ravel()
    ix, iy = np.argsort(x), np.argsort(y)
    xs, ys = x[ix], y[iy]
    rgi = RegularGridInterpolator((ys, xs), np.asarray(z, dtype=np.float64)[np.ix_(iy, ix)], method='linear')
    xn = np.sort(np.atleast_1d(np.asarray(xn, dtype=np.float64)))
    yn = np.sort(np.atleast_1d(np.asarray(yn, dtype=np.float64)))
    yy, xx = np.meshgrid(np.clip(yn, ys[0], ys[-1]), np.clip(xn, xs[0], xs[-1]), indexing='ij')
    return rgi(np.stack([yy.ravel(), xx.ravel()], -1)).reshape(len(yn), len(xn))


def load_topo(GR, data_dir, n_topo_smooth=10):
    """io_initial_conditions.py:185-213"""
    nx, ny, nb = GR.nx, GR.ny, GR.nb
    ii, jj = GR.ii, GR.jj
    f = netcdf_file(os.path.join(data_dir, TOPO_FILE), 'r', mmap=False)
    lon_inp = np.array(f.variables['lon'].data, dtype=np.float64)
    lat_inp = np.array(f.variables['lat'].data, dtype=np.float64)
    hsurf_inp = np.array(f.variables['data'].data[0], dtype=np.float64)
    HSURF = np.full((nx + 2 * nb, ny + 2 * nb, 1), np.nan)
    HSURF[ii, jj, 0] = _bilinear(lon_inp, lat_inp, hsurf_inp,
                                 GR.lon_deg[ii, nb + 1, 0].squeeze(), GR.lat_deg[nb + 1, jj, 0].squeeze()).T
    exchange_BC_host(HSURF, nx, ny)
    HSURF[HSURF < 0] = 0
    exchange_BC_host(HSURF, nx, ny)
    tau = np.full((nx + 2 * nb, ny + 2 * nb, 1), np.nan)
    for j in range(nb, nb + ny):
        tau[:, j, 0] = 0.05 + np.sin(GR.lat_rad[5, j]) ** 2 * (0.15 - 0.05)
    for _ in range(n_topo_smooth):
        HSURF[ii, jj] = HSURF[ii, jj] + (tau[ii, jj] * (HSURF[ii - 1, jj] + HSURF[ii + 1, jj] +
                                                        HSURF[ii, jj - 1] + HSURF[ii, jj + 1] - 4 * HSURF[ii, jj]))
        exchange_BC_host(HSURF, nx, ny)
    return HSURF


def sigma_levels(GR, data_dir, pair_top=10000.):
    """io_initial_conditions.py:253-287 -> sigma_vb (nz+1 values)"""
    HSURF = load_topo(GR, data_dir)
    profile = np.loadtxt(os.path.join(data_dir, PROFILE_FILE))
    zsurf = np.mean(HSURF[GR.ii, GR.jj])
    top_ind = np.argwhere(profile[:, 2] >= pair_top).squeeze()[-1]
    ztop = profile[top_ind, 0] + (profile[top_ind, 2] - pair_top) / (profile[top_ind, 4] * profile[top_ind, 1])
    ks = np.arange(0, GR.nzs)
    z_vb = zsurf + (ztop - zsurf) * (1 - ks / GR.nz) ** 2
    rho_vb = np.interp(z_vb, profile[:, 0], profile[:, 4])
    g_vb = np.interp(z_vb, profile[:, 0], profile[:, 1])
    p_vb = np.zeros(GR.nzs)
    p_vb[0] = pair_top
    for k in range(1, GR.nzs):
        p_vb[k] = p_vb[k - 1] + rho_vb[k] * g_vb[k] * (z_vb[k - 1] - z_vb[k])
    return (p_vb - pair_top) / (p_vb[-1] - pair_top)


def gaussian2D(GR, FIELD, pert, lon0_rad, lat0_rad, lonSig_rad, latSig_rad):
    """io_initial_conditions.py:225-249 on one horizontal slice (modified in place)."""
    dimx, dimy = FIELD.shape
    if dimy == GR.nys + 2 * GR.nb:
        sel, lat, lon = (GR.ii, GR.jjs), GR.lat_js_rad[GR.ii, GR.jjs, 0], GR.lon_js_rad[GR.ii, GR.jjs, 0]
    elif dimx == GR.nxs + 2 * GR.nb:
        sel, lat, lon = (GR.iis, GR.jj), GR.lat_is_rad[GR.iis, GR.jj, 0], GR.lon_is_rad[GR.iis, GR.jj, 0]
    else:
        sel, lat, lon = (GR.ii, GR.jj), GR.lat_rad[GR.ii, GR.jj, 0], GR.lon_rad[GR.ii, GR.jj, 0]
    FIELD[sel] = FIELD[sel] + (pert * np.exp(-np.power(lon - lon0_rad, 2) / (2 * lonSig_rad ** 2)
                                             - np.power(lat - lat0_rad, 2) / (2 * latSig_rad ** 2))).squeeze()
    return FIELD


def initialize_fields(GR, F, data_dir, UWIND_gaussian_pert=10., VWIND_gaussian_pert=10.,
                      gaussian_dlon=np.pi / 10, gaussian_dlat=np.pi / 10):
    """io_initial_conditions.py:31-131 for the fields the dycore owns: HSURF, COLP, POTT, UWIND,
    VWIND (+ zero WWIND, POTTVB), written into F.host with lateral BCs applied."""
    nx, ny, nz = GR.nx, GR.ny, GR.nz
    ii, jj = GR.ii, GR.jj
    pair_top = GR.nl.pair_top
    h = F.host
    h['POTTVB'][:] = 0
    h['WWIND'][:] = 0
    h['HSURF'][...] = load_topo(GR, data_dir)
    profile = np.loadtxt(os.path.join(data_dir, PROFILE_FILE))
    # set_up_profile, :146-180
    PSURF = np.full(h['COLP'].shape, np.nan)
    PSURF[ii, jj, 0] = np.interp(h['HSURF'][ii, jj, 0], profile[:, 0], profile[:, 2])
    h['COLP'][ii, jj, 0] = PSURF[ii, jj, 0] - pair_top
    sig = GR.sigma_vb.reshape(-1)
    PAIRVB = pair_top + sig[None, None, :] * h['COLP'][ii, jj, :]
    PVTFVB = np.power(PAIRVB / 100000., con_kappa)
    PVTF = 1 / (1 + con_kappa) * (PVTFVB[:, :, 1:] * PAIRVB[:, :, 1:] - PVTFVB[:, :, :-1] * PAIRVB[:, :, :-1]) / \
        (PAIRVB[:, :, 1:] - PAIRVB[:, :, :-1])
    PAIR = 100000. * np.power(PVTF, 1 / con_kappa)
    TAIR = interp1d(profile[:, 2], profile[:, 3])(PAIR)
    h['POTT'][ii, jj, :] = TAIR * np.power(100000. / PAIR, con_kappa)
    for k in range(nz):
        h['UWIND'][GR.iis, GR.jj, k] = 0
        h['UWIND'][:, :, k] = gaussian2D(GR, h['UWIND'][:, :, k], UWIND_gaussian_pert, np.pi * 3 / 4, 0,
                                         gaussian_dlon, gaussian_dlat) * (1 - (k + 1) / nz) ** (1 / 2)
        h['VWIND'][:, :, k][GR.ii, GR.jjs] = 0
        h['VWIND'][:, :, k] = gaussian2D(GR, h['VWIND'][:, :, k], VWIND_gaussian_pert, np.pi * 3 / 4, 0,
                                         gaussian_dlon, gaussian_dlat) * (1 - (k + 1) / nz) ** (1 / 2)
    for n in ('COLP', 'UWIND', 'VWIND', 'POTT'):
        exchange_BC_host(h[n], nx, ny)
    return F
