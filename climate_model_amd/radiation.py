"""Radiation scheme of the reference behind its own interface.

Counterpart of `rad_main.Radiation` (rad_main.py:31-205): spectral bins of the Planck sums
(:50-65), the call schedule (`rad_nth_hour`, `launch_radiation_calc`, :68-93) and
`simple_radiation` (:121-205).  The astronomy -- solar zenith angle per column and the current
solar constant, rad_shortwave.py:278-326 -- is 2-D and stays host numpy as in the reference; the
column solves (longwave (2,2)-banded system, Toon shortwave tridiagonal system, flux divergences,
heating rate dPOTTdt_RAD) run on the device: `dycore_radiation`, one thread per column.

Inputs the scheme takes from other packages of the reference (surface scheme: SOILTEMP,
SURFALBEDLW, SURFALBEDSW; microphysics: QC) are fields the caller provides.
"""
import copy
from datetime import timedelta

import numpy as np

from .fields import HIP
from .discretizations import _core_of, _expect

solar_constant_0 = 1365.            # io_constants.py:29
planck_n_lw_bins = 5                # namelist.py:139


def rad_solar_zenith_angle(GR, current_time, SOLZEN):
    """rad_shortwave.py:278-312 (host array, interior columns, in place)."""
    D_J = current_time.timetuple().tm_yday
    Y = current_time.timetuple().tm_year
    if Y >= 2001:
        D_L = np.floor((Y - 2001) / 4)
    else:
        D_L = np.floor((Y - 2000) / 4) - 1
    N_JD = 364.5 + (Y - 2001) * 365 + D_L + D_J
    eps_ob = 23.439 - 0.0000004 * N_JD
    L_M = 280.460 + 0.9856474 * N_JD
    g_M = 357.528 + 0.9856003 * N_JD
    lamb_ec = L_M + 1.915 * np.sin(g_M * np.pi / 180) + 0.020 * np.sin(2 * g_M * np.pi / 180)
    sol_declin = np.arcsin(np.sin(eps_ob / 180 * np.pi) * np.sin(lamb_ec / 180 * np.pi))
    ii, jj = GR.ii, GR.jj
    sec_of_local_noon = GR.lon_rad[ii, jj, 0] / (2 * np.pi) * 86400
    sec_of_day = timedelta(hours=current_time.hour, minutes=current_time.minute,
                           seconds=current_time.second).total_seconds()
    sec_past_local_noon = sec_of_day - sec_of_local_noon
    hour_angle = 2 * np.pi * sec_past_local_noon / 86400
    SOLZEN[ii, jj, 0] = np.arccos(+ np.sin(GR.lat_rad[ii, jj, 0]) * np.sin(sol_declin)
                                  + np.cos(GR.lat_rad[ii, jj, 0]) * np.cos(sol_declin) * np.cos(hour_angle))
    return SOLZEN


def calc_current_solar_constant(current_time):
    """rad_shortwave.py:315-326"""
    n_days_in_year = 365
    D_J = current_time.timetuple().tm_yday
    theta_J = 2 * np.pi * D_J / n_days_in_year
    earth_sun_dist = (1.00011 + 0.034221 * np.cos(theta_J) + 0.00128 * np.sin(theta_J)
                      + 0.000719 * np.cos(2 * theta_J) + 0.000077 * np.sin(2 * theta_J))
    return solar_constant_0 * earth_sun_dist


class Radiation:

    fields_radiation = ['PHIVB', 'SOLZEN', 'MYSUN', 'SWINTOA', 'TAIR',
                        'RHO', 'SOILTEMP', 'SURFALBEDLW', 'SURFALBEDSW',
                        'QC', 'LWFLXDO', 'LWFLXUP', 'SWDIFFLXDO',
                        'SWDIRFLXDO', 'SWFLXUP', 'SWFLXDO',
                        'LWFLXNET', 'SWFLXNET', 'SWFLXDIV', 'LWFLXDIV',
                        'TOTFLXDIV', 'dPOTTdt_RAD']

    def __init__(self, GR, rad_nth_hour=3.9, target=HIP):
        if target != HIP:
            raise ValueError("the only compute target of this package is 'HIP'")
        self.target = target
        self.done = 0
        self.rad_nth_hour = rad_nth_hour
        self.rad_nth_ts = max(1, int(self.rad_nth_hour * 3600 / GR.dt))
        # Planck emission bins: rad_main.py:50-65
        nu0, nu1, nnu = 50. * 100, 2500. * 100, planck_n_lw_bins
        dnu = (nu1 - nu0) / nnu
        nus = np.arange(nu0, nu1 + 1, dnu).astype(np.float64)
        nus_center = nus[:-1] + dnu / 2
        lambdas = 1. / nus
        self.planck_lambdas_center = 1. / nus_center
        self.planck_dlambdas = np.diff(lambdas)
        self.solar_constant = None

    def launch_radiation_calc(self, GR, F):
        """rad_main.py:68-93, synchronous branch."""
        if GR.ts % self.rad_nth_ts == 0:
            self.calc_radiation(GR, F)

    def calc_radiation(self, GR, F):
        GR.timer.start('rad')
        self.simple_radiation(GR, host=F.host, **F.get(self.fields_radiation, target=self.target))
        GR.timer.stop('rad')
        self.done = 1

    def simple_radiation(self, GR, PHIVB, SOLZEN, MYSUN, SWINTOA, TAIR,
                         RHO, SOILTEMP, SURFALBEDLW, SURFALBEDSW, QC, LWFLXDO,
                         LWFLXUP, SWDIFFLXDO, SWDIRFLXDO, SWFLXUP, SWFLXDO,
                         LWFLXNET, SWFLXNET, SWFLXDIV, LWFLXDIV, TOTFLXDIV,
                         dPOTTdt_RAD, host=None, current_GMT=None):
        """rad_main.py:121-205.  The three astronomical 2-D fields are computed on the host
        (`host` = the ModelFields host dict that receives them) and uploaded."""
        _expect(PHIVB=PHIVB, SOLZEN=SOLZEN, MYSUN=MYSUN, SWINTOA=SWINTOA, TAIR=TAIR, RHO=RHO, SOILTEMP=SOILTEMP,
                SURFALBEDLW=SURFALBEDLW, SURFALBEDSW=SURFALBEDSW, QC=QC, LWFLXDO=LWFLXDO, LWFLXUP=LWFLXUP,
                SWDIFFLXDO=SWDIFFLXDO, SWDIRFLXDO=SWDIRFLXDO, SWFLXUP=SWFLXUP, SWFLXDO=SWFLXDO,
                LWFLXNET=LWFLXNET, SWFLXNET=SWFLXNET, SWFLXDIV=SWFLXDIV, LWFLXDIV=LWFLXDIV,
                TOTFLXDIV=TOTFLXDIV, dPOTTdt_RAD=dPOTTdt_RAD)
        core = _core_of(PHIVB, dPOTTdt_RAD)
        current_GMT = copy.copy(current_GMT if current_GMT is not None else GR.GMT)
        shape = (GR.nx + 2, GR.ny + 2, 1)
        h = host if host is not None else {}
        solzen = h['SOLZEN'] if 'SOLZEN' in h else np.full(shape, np.nan)
        solzen = rad_solar_zenith_angle(GR, current_GMT, solzen)
        mysun = np.cos(solzen)                                   # negative at night: no shortwave there
        self.solar_constant = calc_current_solar_constant(current_GMT)
        swintoa = self.solar_constant * np.cos(solzen)
        if host is not None:
            host['SOLZEN'], host['MYSUN'], host['SWINTOA'] = solzen, mysun, swintoa
        core.upload('SOLZEN', solzen)
        core.upload('MYSUN', mysun)
        core.upload('SWINTOA', swintoa)
        core.radiation(self.solar_constant, self.planck_lambdas_center, self.planck_dlambdas)


def band_radiation(RAD, GR, layout, rank, core, current_GMT=None):
    """The radiation call of one latitude band (climate_model_amd.bands): the scheme is column-local,
    so a band needs no neighbour rows -- the astronomy is evaluated on the global grid `GR` and this
    band's rows (halo rows included) are uploaded to its handle."""
    gmt = copy.copy(current_GMT if current_GMT is not None else GR.GMT)
    solzen = rad_solar_zenith_angle(GR, gmt, np.full((GR.nx + 2, GR.ny + 2, 1), np.nan))
    RAD.solar_constant = calc_current_solar_constant(gmt)
    core.upload('SOLZEN', layout.slice_rows(solzen, rank, 0))
    core.upload('MYSUN', layout.slice_rows(np.cos(solzen), rank, 0))
    core.upload('SWINTOA', layout.slice_rows(RAD.solar_constant * np.cos(solzen), rank, 0))
    core.radiation(RAD.solar_constant, RAD.planck_lambdas_center, RAD.planck_dlambdas)
