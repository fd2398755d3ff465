"""Deterministic, file-free initial state for benchmarks and property tests.

Recipe of SURVEY.md 8(d): Gaussian-hill topography, hydrostatic column pressure,
isothermal (288 K) potential-temperature profile, the reference's Gaussian wind bump
(io_initial_conditions.py:67-78: 10 m/s at lon 135 deg / lat 0, sigma = pi/10, scaled by
sqrt(1-(k+1)/nz)) plus seeded uniform noise so that no field is constant.  The
physics-produced inputs (KMOM, KHEAT, surface fluxes) stay zero: dry dycore.
"""
import numpy as np

from .constants import con_kappa
from .grid import exchange_BC_host


def _gauss(lon, lat, lon0, lat0, slon, slat):
    return np.exp(-(lon - lon0) ** 2 / (2 * slon ** 2) - (lat - lat0) ** 2 / (2 * slat ** 2))


def _noise(i, j, k, salt):
    """Uniform(-1,1) noise as a pure function of the GLOBAL cell index, so that a latitude
    band generates exactly the values the whole grid would (integer hash, 2^-32 steps)."""
    with np.errstate(over='ignore'):
        x = (np.asarray(i, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
             + np.asarray(j, dtype=np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F)
             + np.uint64((k + 1) * 0x165667B19E3779F9 % (1 << 64)) + np.uint64(salt))
        x ^= x >> np.uint64(31)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(29)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(32)
    return (x >> np.uint64(32)).astype(np.float64) / 2147483648.0 - 1.0


def initialize_fields(GR, F, seed=3, noise_wind=0.1, noise_pott=0.1, j_offset=0):
    """Fill F.host with the synthetic state (in place) and apply the lateral BCs.
    `j_offset` shifts local rows to global rows for the noise of a latitude band."""
    nx, ny, nz = GR.nx, GR.ny, GR.nz
    ii, jj, iis, jjs = GR.ii, GR.jj, GR.iis, GR.jjs
    pair_top = GR.nl.pair_top
    h = F.host

    lon, lat = GR.lon_rad[ii, jj, 0], GR.lat_rad[ii, jj, 0]
    hs = (3000. * _gauss(lon, lat, np.pi * 0.5, 0.6, 0.35, 0.25) +
          1500. * _gauss(lon, lat, np.pi * 1.4, -0.4, 0.5, 0.3) +
          500. * _gauss(lon, lat, np.pi * 1.9, 0.9, 0.3, 0.2))
    h['HSURF'][ii, jj, 0] = hs
    exchange_BC_host(h['HSURF'], nx, ny)
    h['COLP'][ii, jj, 0] = 101325. * np.exp(-hs / 8000.) - pair_top
    exchange_BC_host(h['COLP'], nx, ny)

    sig = GR.sigma_vb.reshape(-1)
    sig_mid = 0.5 * (sig[:-1] + sig[1:])
    bump_u = 10. * _gauss(GR.lon_is_rad[iis, jj, 0], GR.lat_is_rad[iis, jj, 0], np.pi * 3 / 4, 0., np.pi / 10, np.pi / 10)
    bump_v = 10. * _gauss(GR.lon_js_rad[ii, jjs, 0], GR.lat_js_rad[ii, jjs, 0], np.pi * 3 / 4, 0., np.pi / 10, np.pi / 10)
    colp = h['COLP'][ii, jj, 0]
    for k in range(nz):
        vert = (1 - (k + 1) / nz) ** 0.5
        h['UWIND'][iis, jj, k] = bump_u * vert + noise_wind * _noise(iis, jj + j_offset, k, seed)
        h['VWIND'][ii, jjs, k] = bump_v * vert + noise_wind * _noise(ii, jjs + j_offset, k, seed + 101)
        p_mid = pair_top + sig_mid[k] * colp
        h['POTT'][ii, jj, k] = 288. * (100000. / p_mid) ** con_kappa + noise_pott * _noise(ii, jj + j_offset, k, seed + 202)
    for n in ('UWIND', 'VWIND', 'POTT'):
        exchange_BC_host(h[n], nx, ny)
    # x-staggered arrays carry one more periodic column than Grid.exchange_BC fills
    h['UWIND'][nx + 2] = h['UWIND'][2]
    return F


def initialize_moisture(GR, F):
    """A smooth specific-humidity / cloud-water field with structure in every direction (QV, QC on
    the global grid, lateral BCs applied) and a constant surface latent heat flux: the inputs of
    dyn_moist for synthetic runs (the reference starts from 60 % relative humidity, mic_main.py:63-71)."""
    from .grid import exchange_BC_host
    ii, jj = np.meshgrid(np.arange(GR.nx + 2), np.arange(GR.ny + 2), indexing='ij')
    for k in range(GR.nz):
        F.host['QV'][:, :, k] = 0.004 * (k + 1) / GR.nz * (1.2 + np.sin(2 * np.pi * ii / GR.nx) * np.cos(np.pi * jj / GR.ny))
        F.host['QC'][:, :, k] = 1e-5 * (1.1 + np.cos(4 * np.pi * ii / GR.nx)) * (k + 1) / GR.nz
    exchange_BC_host(F.host['QV'], GR.nx, GR.ny)
    exchange_BC_host(F.host['QC'], GR.nx, GR.ny)
    F.host['SLHFLX'][...] = 30.
    return F


def stable_dt(GR, safety=0.5):
    """A CFL-safe time step for grids where the reference's int() rule gives 0."""
    return safety * GR.nl.CFL * float(np.nanmin(GR.dx)) / 400.
