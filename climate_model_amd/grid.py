"""Computational grid of the dycore: extents, metric terms, sigma levels, time step.

Counterpart of the reference's `main_grid.Grid` (main_grid.py:80-314) restricted to
what the Matsuno step consumes: the `GRF` dict of grid constants with the
reference's names and shapes, `dt`, a `timer`, and the extents.  Metric terms use
the reference's closed forms (main_grid.py:166-242).  Sigma levels: the reference
derives them from a standard-atmosphere table and its topography file
(io_initial_conditions.py:253-287); without those inputs `Grid` uses the analytic
stand-in sigma_vb = (k/nz)**1.5 unless `sigma_vb` is passed in.
"""
import numpy as np

from .constants import con_rE, con_omega
from .fields import CPU, HIP
from .namelist import Namelist
from .utilities import Timer


def exchange_BC_host(FIELD, nx, ny):
    """Host-side lateral BC for initialisation (main_grid.py:317-357; like the reference's
    Grid.exchange_BC it does not fill column nx+2 of x-staggered arrays)."""
    nxs, nys = nx + 1, ny + 1
    fnx, fny = FIELD.shape[0], FIELD.shape[1]
    if fnx == nxs + 2:
        FIELD[0] = FIELD[nxs - 1]
        FIELD[nxs] = FIELD[1]
    else:
        FIELD[0] = FIELD[nx]
        FIELD[nx + 1] = FIELD[1]
    if fny == nys + 2:
        for j in (0, 1, nys, nys + 1):
            FIELD[:, j] = 0.
    else:
        FIELD[:, 0] = FIELD[:, 1]
        FIELD[:, ny + 1] = FIELD[:, ny]
    return FIELD


class Grid:

    def __init__(self, nl=None, sigma_vb=None):
        nl = nl or Namelist()
        self.nl = nl
        wp = np.float64
        self.nb = nb = nl.nb
        self.nz = int(nl.nz)
        self.nzs = self.nz + 1
        self.nx = int((nl.lon1_deg - nl.lon0_deg) / nl.dlon_deg)
        self.nxs = self.nx + 1
        self.ny = int((nl.lat1_deg - nl.lat0_deg) / nl.dlat_deg)
        self.nys = self.ny + 1
        nx, ny, nz, nxs, nys = self.nx, self.ny, self.nz, self.nxs, self.nys
        self.dlon_deg, self.dlat_deg = nl.dlon_deg, nl.dlat_deg
        self.dlon_rad_1D = nl.dlon_deg / 180 * np.pi
        self.dlat_rad_1D = nl.dlat_deg / 180 * np.pi
        self.i = np.arange(nb, nx + nb)
        self.i_s = np.arange(nb, nxs + nb)
        self.j = np.arange(nb, ny + nb)
        self.js = np.arange(nb, nys + nb)
        self.ii, self.jj = np.ix_(self.i, self.j)
        self.iis, self.jjs = np.ix_(self.i_s, self.js)

        def full(stgx, stgy, val=np.nan):
            return np.full((nx + 2 * nb + stgx, ny + 2 * nb + stgy, 1), val, dtype=wp)

        # latitudes / longitudes of mass, u and v points (main_grid.py:132-189)
        self.lon_deg, self.lat_deg = full(0, 0), full(0, 0)
        self.lon_is_deg, self.lat_is_deg = full(1, 0), full(1, 0)
        self.lon_js_deg, self.lat_js_deg = full(0, 1), full(0, 1)
        self.dlon_rad = full(0, 1, self.dlon_rad_1D)
        self.dlat_rad = full(1, 0, self.dlat_rad_1D)
        self.lon_deg[self.ii, self.jj, 0] = nl.lon0_deg + (self.ii - nb + 0.5) * nl.dlon_deg
        self.lon_is_deg[self.iis, self.jj, 0] = nl.lon0_deg + (self.iis - nb) * nl.dlon_deg
        self.lon_js_deg[self.ii, self.jjs, 0] = nl.lon0_deg + (self.ii - nb + 0.5) * nl.dlon_deg
        self.lat_deg[self.ii, self.jj, 0] = nl.lat0_deg + (self.jj - nb + 0.5) * nl.dlat_deg
        self.lat_js_deg[self.ii, self.jjs, 0] = nl.lat0_deg + (self.jjs - nb) * nl.dlat_deg
        self.lat_is_deg[self.iis, self.jj, 0] = nl.lat0_deg + (self.jj - nb + 0.5) * nl.dlat_deg
        self.lon_rad = self.lon_deg / 180 * np.pi
        self.lat_rad = self.lat_deg / 180 * np.pi
        self.lon_is_rad = self.lon_is_deg / 180 * np.pi
        self.lat_is_rad = self.lat_is_deg / 180 * np.pi
        self.lon_js_rad = self.lon_js_deg / 180 * np.pi
        self.lat_js_rad = self.lat_js_deg / 180 * np.pi

        # grid spacing and cell area (main_grid.py:192-227)
        self.dx, self.dxjs, self.dyis = full(0, 0), full(0, 1), full(1, 0)
        self.dx[self.ii, self.jj, 0] = np.cos(self.lat_rad[self.ii, self.jj, 0]) * self.dlon_rad_1D * con_rE
        self.dxjs[self.ii, self.jjs, 0] = np.cos(self.lat_js_rad[self.ii, self.jjs, 0]) * self.dlon_rad_1D * con_rE
        self.dyis[self.iis, self.jj, 0] = self.dlat_rad_1D * con_rE
        for f in (self.dx, self.dxjs, self.dyis):
            exchange_BC_host(f, nx, ny)
        self.A = full(0, 0)
        self.A[self.ii, self.jj, 0] = (np.cos(self.lat_rad[self.ii, self.jj, 0]) *
                                       self.dlon_rad_1D * self.dlat_rad_1D * con_rE ** 2)
        exchange_BC_host(self.A, nx, ny)

        # Coriolis parameter (main_grid.py:229-240)
        self.corf, self.corf_is = full(0, 0), full(1, 0)
        self.corf[self.ii, self.jj, 0] = 2 * con_omega * np.sin(self.lat_rad[self.ii, self.jj, 0])
        self.corf_is[self.iis, self.jj, 0] = 2 * con_omega * np.sin(self.lat_is_rad[self.iis, self.jj, 0])

        # sigma levels
        if sigma_vb is None:
            sigma_vb = (np.arange(self.nzs) / nz) ** 1.5
        sigma_vb = np.asarray(sigma_vb, dtype=wp).reshape(-1)
        assert sigma_vb.size == self.nzs
        self.sigma_vb = sigma_vb.reshape(1, 1, self.nzs).copy()
        self.dsigma = np.diff(sigma_vb).reshape(1, 1, nz)

        # time step (main_grid.py:258-265)
        if nl.dt is None:
            mindx = np.nanmin(self.dx)
            self.dt = int(nl.CFL * mindx / 400)
            if self.dt < 1:
                # the reference's rule gives 0 below ~1 deg with the poles included (its own
                # `while ... % dt` loop would divide by zero); use half the CFL bound instead
                self.dt = 0.5 * nl.CFL * mindx / 400
            else:
                while nl.i_out_nth_hour * 3600 % self.dt > 0:        # main_grid.py:262-263
                    self.dt -= 1
        else:
            self.dt = nl.dt
        # main_grid.py:264-270
        self.i_sim_n_days, self.i_out_nth_hour, self.i_restart_nth_day = nl.i_sim_n_days, nl.i_out_nth_hour, nl.i_restart_nth_day
        self.nts = nl.i_sim_n_days * 3600 * 24 / self.dt
        self.i_out_nth_ts = int(nl.i_out_nth_hour * 3600 / self.dt)
        self.i_restart_nth_ts = int(nl.i_restart_nth_day * 24 / nl.i_out_nth_hour * self.i_out_nth_ts)
        self.nc_output_count = 0
        self.ts = 0
        self.GMT = nl.GMT_initialization        # main_grid.py:273; advanced by the solver loop
        self.timer = Timer()

        # numerical diffusion per level (main_grid.py:275-294)
        k = np.arange(nz)
        self.UVFLX_dif_coef = np.zeros((1, 1, nz), dtype=wp)
        self.POTT_dif_coef = np.zeros((1, 1, nz), dtype=wp)
        self.UVFLX_dif_coef[0, 0, k] = nl.UVFLX_dif_coef * np.exp(-0.0 * (nz - k - 1) / nz)
        self.POTT_dif_coef[0, 0, k] = nl.POTT_dif_coef * np.exp(-1.5 * (nz - k - 1) / nz)
        self.moist_dif_coef = np.zeros((1, 1, nz), dtype=wp)
        self.moist_dif_coef[0, 0, k] = nl.moist_dif_coef * np.exp(-1.5 * (nz - k - 1) / nz)
        self._build_GRF()

    GRID_FIELD_NAMES = ['corf', 'corf_is', 'A', 'sigma_vb', 'dsigma', 'dxjs', 'dyis',
                        'lat_rad', 'lat_is_rad', 'dlat_rad', 'dlon_rad',
                        'POTT_dif_coef', 'UVFLX_dif_coef', 'moist_dif_coef']

    def _build_GRF(self):
        """main_grid.py:299-314.  The device copy lives inside the HIP handle
        (dycore_set_grid); both targets expose the same host arrays here."""
        grf = {name: getattr(self, name) for name in self.GRID_FIELD_NAMES}
        self.GRF = {CPU: grf, HIP: grf}

    @classmethod
    def from_arrays(cls, nx, ny, nz, grf, dt, pair_top=10000., i_turb=0, i_moist=0):
        """A Grid around externally supplied grid constants (golden fixtures, restarts)."""
        self = cls.__new__(cls)
        self.nl = Namelist(nz=nz, dlat_deg=180. / ny, dlon_deg=360. / nx, pair_top=pair_top,
                           i_UVFLX_vert_turb=i_turb, i_POTT_vert_turb=i_turb, i_moist_main_switch=i_moist, dt=dt)
        self.nb = 1
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        self.nxs, self.nys, self.nzs = self.nx + 1, self.ny + 1, self.nz + 1
        self.i = np.arange(1, self.nx + 1)
        self.i_s = np.arange(1, self.nxs + 1)
        self.j = np.arange(1, self.ny + 1)
        self.js = np.arange(1, self.nys + 1)
        self.ii, self.jj = np.ix_(self.i, self.j)
        self.iis, self.jjs = np.ix_(self.i_s, self.js)
        for name in cls.GRID_FIELD_NAMES:
            src = name if name in grf else 'POTT_dif_coef'       # moist_dif_coef defaults to POTT's
            setattr(self, name, np.array(grf[src], dtype=np.float64))
        self.dt = dt
        self.ts = 0
        self.GMT = self.nl.GMT_initialization
        self.timer = Timer()
        self._build_GRF()
        return self
