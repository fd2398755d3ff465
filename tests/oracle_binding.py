"""ctypes binding of oracle/libdycore_oracle.so -- test infrastructure only.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by the climate_model_amd package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from climate_model_amd.fields import FIELD_TABLE, field_shape

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')
LIB_PATH = os.path.join(ORACLE_DIR, 'libdycore_oracle.so')
LIB_PATH_F32 = os.path.join(ORACLE_DIR, 'libdycore_oracle_f32.so')     # working_precision = 'float32'

_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)


class OracleCfg(C.Structure):
    _fields_ = [('nx', C.c_int), ('ny', C.c_int), ('nz', C.c_int),
                ('i_turb', C.c_int), ('i_radiation', C.c_int), ('nthreads', C.c_int),
                ('pair_top', C.c_double), ('i_moist', C.c_int), ('i_moist_clip', C.c_int), ('term_off', C.c_int)]


GRID_2D = ['A', 'dxjs', 'dyis', 'corf', 'corf_is', 'lat_rad', 'lat_is_rad', 'dlon_rad', 'dlat_rad']
GRID_1D = ['sigma_vb', 'dsigma', 'UVFLX_dif_coef', 'POTT_dif_coef', 'moist_dif_coef']


class OracleGrid(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in GRID_2D + GRID_1D]


ORACLE_FIELDS = ['COLP', 'COLP_OLD', 'COLP_NEW', 'dCOLPdt', 'HSURF',
                 'UWIND', 'UWIND_OLD', 'VWIND', 'VWIND_OLD', 'WWIND',
                 'POTT', 'POTT_OLD', 'POTTVB',
                 'UFLX', 'VFLX', 'FLXDIV',
                 'BFLX', 'CFLX', 'DFLX', 'EFLX', 'RFLX', 'QFLX', 'SFLX', 'TFLX',
                 'WWIND_UWIND', 'WWIND_VWIND', 'KMOM_dUWINDdz', 'KMOM_dVWINDdz',
                 'dUFLXdt', 'dVFLXdt', 'dPOTTdt',
                 'dUFLXdt_TURB', 'dVFLXdt_TURB', 'dPOTTdt_TURB',
                 'PVTF', 'PVTFVB', 'PHI', 'PHIVB',
                 'RHO', 'RHOVB', 'KMOM', 'KHEAT',
                 'SSHFLX', 'SMOMXFLX', 'SMOMYFLX', 'dPOTTdt_RAD',
                 'QV', 'QV_OLD', 'QC', 'QC_OLD', 'dQVdt', 'dQCdt', 'dQVdt_TURB', 'SLHFLX',
                 'TAIR', 'TAIRVB', 'PAIR', 'PAIRVB', 'WINDX', 'WINDY', 'WIND']


class OracleFields(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ORACLE_FIELDS]


_lib = None
_lib_f32 = None


def build_oracle(force=False):
    src = [os.path.join(ORACLE_DIR, f) for f in ('dycore_oracle.c', 'dycore_oracle.h', 'Makefile')]
    stale = any(not os.path.exists(L) or any(os.path.getmtime(s) > os.path.getmtime(L) for s in src)
                for L in (LIB_PATH, LIB_PATH_F32))
    if force or stale:
        subprocess.run(['make', '-C', ORACLE_DIR, '-s'], check=True)
    return LIB_PATH


def load_oracle(dtype=np.float64):
    global _lib, _lib_f32
    f32 = np.dtype(dtype) == np.float32
    if (_lib_f32 if f32 else _lib) is None:
        build_oracle()
        lib = C.CDLL(LIB_PATH_F32 if f32 else LIB_PATH)
        cp, gp, fp = C.POINTER(OracleCfg), C.POINTER(OracleGrid), C.POINTER(OracleFields)
        lib.oracle_exchange_bc.argtypes = [cp, C.c_void_p, C.c_int, C.c_int, C.c_int]
        for n in ('oracle_continuity', 'oracle_make_timestep', 'oracle_stage_continuity',
                  'oracle_stage_euler_forward'):
            getattr(lib, n).argtypes = [cp, gp, fp, C.c_double]
        for n in ('oracle_uvflx_prepare', 'oracle_uflx_tendency', 'oracle_vflx_tendency',
                  'oracle_pott_tendency', 'oracle_diag_pvtf', 'oracle_stage_momentum',
                  'oracle_stage_temperature', 'oracle_stage_primary_diag', 'oracle_stage_moisture',
                  'oracle_moist_tendency'):
            getattr(lib, n).argtypes = [cp, gp, fp]
        for n in ('oracle_diag_phi', 'oracle_diag_pottvb', 'oracle_diag_rho', 'oracle_diag_secondary',
                  'oracle_compute_turbulence'):
            getattr(lib, n).argtypes = [cp, fp]
        lib.oracle_step_matsuno.argtypes = [cp, gp, fp, C.c_double, C.c_int, C.c_int]
        if f32:
            _lib_f32 = lib
        else:
            _lib = lib
    return _lib_f32 if f32 else _lib


class Oracle:
    """The CPU restatement bound to one set of numpy arrays (reference layout)."""

    def __init__(self, nx, ny, nz, grid, fields=None, pair_top=10000., i_turb=1,
                 i_radiation=0, nthreads=0, fill_nan=True, i_moist=0, i_moist_clip=0, dtype=np.float64, term_off=0):
        self.dtype = np.dtype(dtype)
        self.lib = load_oracle(self.dtype)
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        if not nthreads:        # a GPU box shows every core of its host; its share is 16
            nthreads = min(16, len(os.sched_getaffinity(0)))
        self.cfg = OracleCfg(self.nx, self.ny, self.nz, int(i_turb), int(i_radiation),
                             int(nthreads), float(pair_top), int(i_moist), int(i_moist_clip), int(term_off))
        self.G = {}
        self.grid = OracleGrid()
        for n in GRID_2D + GRID_1D:
            if n == 'moist_dif_coef' and n not in grid:      # dry callers: same vertical profile as POTT
                grid = dict(grid, moist_dif_coef=grid['POTT_dif_coef'])
            a = np.ascontiguousarray(np.asarray(grid[n], dtype=self.dtype))
            self.G[n] = a
            setattr(self.grid, n, a.ctypes.data)
        self.F = {}
        self.fields = OracleFields()
        for n in ORACLE_FIELDS:
            shp = field_shape(n, self.nx, self.ny, self.nz)
            if fields is not None and n in fields:
                a = np.array(fields[n], dtype=self.dtype, order='C')
                assert a.shape == shp, (n, a.shape, shp)
            elif fill_nan:
                a = np.full(shp, np.nan, dtype=self.dtype)
            else:               # benchmark use: untouched pages cost nothing
                a = np.empty(shp, dtype=self.dtype)
            self.F[n] = a
            setattr(self.fields, n, a.ctypes.data)

    def set(self, name, value):
        self.F[name][...] = value

    def _c(self):
        return C.byref(self.cfg), C.byref(self.grid), C.byref(self.fields)

    def exchange_bc(self, name):
        stgx, stgy, _ = FIELD_TABLE[name]
        a = self.F[name]
        self.lib.oracle_exchange_bc(C.byref(self.cfg), a.ctypes.data, stgx, stgy, a.shape[2])

    def continuity(self, dt):
        c, g, f = self._c(); self.lib.oracle_stage_continuity(c, g, f, float(dt))

    def momentum(self):
        c, g, f = self._c(); self.lib.oracle_stage_momentum(c, g, f)

    def temperature(self):
        c, g, f = self._c(); self.lib.oracle_stage_temperature(c, g, f)

    def moisture(self):
        c, g, f = self._c(); self.lib.oracle_stage_moisture(c, g, f)

    def euler_forward(self, dt):
        c, g, f = self._c(); self.lib.oracle_stage_euler_forward(c, g, f, float(dt))

    def primary_diag(self):
        c, g, f = self._c(); self.lib.oracle_stage_primary_diag(c, g, f)

    def diag_rho(self):
        c, g, f = self._c(); self.lib.oracle_diag_rho(c, f)

    def secondary_diag(self):
        c, g, f = self._c(); self.lib.oracle_diag_secondary(c, f)

    def compute_turbulence(self):
        c, g, f = self._c(); self.lib.oracle_compute_turbulence(c, f)

    def step_matsuno(self, dt, nsteps=1, with_rho=True):
        c, g, f = self._c()
        self.lib.oracle_step_matsuno(c, g, f, float(dt), int(nsteps), int(bool(with_rho)))


def oracle_from_golden(z, tag=None, i_turb=1, i_moist=0, dtype=np.float64, term_off=0):
    """Build an Oracle from a golden .npz: grid constants + either the initial
    conditions (tag None) or the full captured state `tag` (e.g. 'st00_pre')."""
    nx, ny, nz = int(z['meta/nx']), int(z['meta/ny']), int(z['meta/nz'])
    grid = {n: z['grid/' + n] for n in GRID_2D + GRID_1D}
    fields = {}
    prefix = (tag + '/') if tag else 'ic/'
    for n in ORACLE_FIELDS:
        if prefix + n in z:
            fields[n] = z[prefix + n]
    o = Oracle(nx, ny, nz, grid, fields, pair_top=float(z['meta/pair_top']), i_turb=i_turb, i_moist=i_moist, dtype=dtype, term_off=term_off)
    return o


RAD_INPUTS = ['PHIVB', 'TAIR', 'RHO', 'SOILTEMP', 'SURFALBEDLW', 'SURFALBEDSW', 'QC', 'MYSUN', 'SWINTOA']
RAD_OUTPUTS = ['LWFLXDO', 'LWFLXUP', 'SWDIFFLXDO', 'SWDIRFLXDO', 'SWFLXUP', 'SWFLXDO', 'LWFLXNET', 'SWFLXNET',
               'SWFLXDIV', 'LWFLXDIV', 'TOTFLXDIV', 'dPOTTdt_RAD']
RAD_NZS = ('LWFLXDO', 'LWFLXUP', 'SWDIFFLXDO', 'SWDIRFLXDO', 'SWFLXUP', 'SWFLXDO', 'LWFLXNET', 'SWFLXNET')


def oracle_radiation(nx, ny, nz, inputs, solar_constant, lambdas_center, dlambdas):
    """radiation_oracle.c on reference-layout arrays; returns the 12 output arrays (NaN outside the
    interior columns)."""
    lib = load_oracle()
    dp = C.POINTER(C.c_double)
    lib.oracle_radiation.restype = None
    lib.oracle_radiation.argtypes = [C.c_int] * 3 + [dp] * 9 + [C.c_double, C.c_int, dp, dp] + [dp] * 12
    a = {n: np.ascontiguousarray(inputs[n], dtype=np.float64) for n in RAD_INPUTS}
    lam = np.ascontiguousarray(lambdas_center, dtype=np.float64)
    dlam = np.ascontiguousarray(dlambdas, dtype=np.float64)
    out = {n: np.full((nx + 2, ny + 2, nz + 1 if n in RAD_NZS else nz), np.nan) for n in RAD_OUTPUTS}
    lib.oracle_radiation(nx, ny, nz, *[a[n].ctypes.data_as(dp) for n in RAD_INPUTS], float(solar_constant),
                         lam.size, lam.ctypes.data_as(dp), dlam.ctypes.data_as(dp),
                         *[out[n].ctypes.data_as(dp) for n in RAD_OUTPUTS])
    return out
