#!/usr/bin/env python3
"""
Golden-vector generator for the Matsuno dycore step.   TEST INFRASTRUCTURE ONLY.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  It imports the reference's own Python modules
(dyn_matsuno.step_matsuno and everything below it) and executes their scalar
formulae as plain IEEE-754 Python, then writes small .npz fixtures under
tests/golden/.  Nothing of the reference's source text is copied: the fixtures
hold inputs and outputs only.

Why stand-ins are needed: the reference decorates its
pure-Python scalar functions with numba.njit, and numba is not installed in
this image.  The stand-ins written below (to a temp dir, at run time) are
  * `numba`      : njit/jit = identity decorator, prange = range (the decorated
                   functions are unchanged reference code, executed by CPython);
                   ndarray arguments are viewed through a subclass whose scalar
                   __getitem__ reproduces numba's unchecked indexing: an
                   out-of-range index reads the element at that flat offset of the
                   C-contiguous allocation (NaN beyond it).  The dry path never
                   uses such a value; dyn_moist multiplies one by WWIND[nz] = 0
                   (dyn_moist.py:216, QV[i,j,k+1] at k = nz-1).  A function that
                   returns a never-assigned local yields 0
                   (dyn_functions.py:310/377, levels never consumed).
  * `netCDF4`    : Dataset reader over scipy.io.netcdf_file (topography input).
  * `scipy.interpolate.interp2d` : bilinear interpolation (removed from SciPy).
  * `bin.rad_longwave_cython`    : empty (radiation is off).
  * `namelist`   : the reference's namelist.py is executed, then the run-mode
                   values are overridden (CPU mode, no restart, physics off,
                   grid extents); the dlat->UVFLX_dif_coef table rule of
                   namelist.py:286-316 is re-applied for the overridden dlat.
Consequently the golden vectors are "the reference's formulae evaluated in
IEEE fp64 by CPython/libm"; true numba bit patterns are unobtainable offline
(numba sums sequentially where numpy sums pairwise, LLVM pow vs libm pow).
Tolerances in tests/ are stated accordingly.

Usage:
    python oracle/gen_golden.py            # regenerate every fixture
    python oracle/gen_golden.py --case g10x4_dry   # one case (subprocess body)
"""
import argparse
import os
import subprocess
import sys
import tempfile
import textwrap

import numpy as np

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(os.path.dirname(HERE), 'tests', 'golden')

# name -> (dlat=dlon [deg], nz, n_steps, snapshot steps, stage capture, turb variant)
CASES = {
    'g10x4_dry':  dict(d=10, nz=4, steps=10, snaps=(1, 2, 10), stages=True,  turb=False),
    'g10x4_turb': dict(d=10, nz=4, steps=1,  snaps=(1,),       stages=True,  turb=True),
    'g4x8_dry':   dict(d=4,  nz=8, steps=10, snaps=(1, 10),    stages=False, turb=False),
    # working_precision = 'float32' (namelist.py:225-226): the reference's arrays and wp() constants in float32
    'g10x4_dry_f32': dict(d=10, nz=4, steps=10, snaps=(1, 2, 10), stages=True, turb=False, wp='float32'),
    # turbulence scheme of the reference switched on (turb_main.Turbulence): KMOM/KHEAT are
    # produced by compute_turbulence from the full secondary diagnostics every step
    'g10x4_phys': dict(d=10, nz=4, steps=5,  snaps=(1, 5),     stages=True,  turb=False, turbulence=True),
    # radiation columns (rad_main.Radiation.simple_radiation) on the model state after two steps
    'g10x4_rad':  dict(d=10, nz=4, steps=2,  snaps=(),         stages=False, turb=False, radiation=(2001, 3, 21, 9, 30, 0)),
    'g4x8_rad':   dict(d=4,  nz=8, steps=2,  snaps=(),         stages=False, turb=False, radiation=(1999, 12, 30, 22, 15, 40)),
    # the reference's "testsuite equality" preset (namelist.py:246-261): 5 deg, +-80 deg, 8 levels, 0.72 days,
    # history every 8 h; turbulence + radiation + moisture on, surface scheme / microphysics off (their
    # products SOILTEMP, albedos, surface fluxes, QC sources are fixed inputs).  Runs the loop of solver.py:82-226
    # up to the second history file (out0002 = the file testsuite.py compares).
    # per-term dynamics switches of namelist.py:62-88 set to 0 (consumed at dyn_UFLX.py:117-193, dyn_VFLX.py:115-192,
    # dyn_POTT.py:72-108, dyn_moist.py:68-115, dyn_UVFLX_prepare.py:260-356): between the two cases every switch that
    # travels in DycoreConfig.term_off is off once
    'g10x4_termsA': dict(d=10, nz=4, steps=3, snaps=(1, 3), stages=True, turb=False,
                         terms=dict(i_UVFLX_hor_adv=0, i_UVFLX_num_dif=0, i_POTT_vert_adv=0, i_POTT_num_dif=0, i_moist_hor_adv=0)),
    'g10x4_termsB': dict(d=10, nz=4, steps=3, snaps=(1, 3), stages=True, turb=False,
                         terms=dict(i_UVFLX_vert_adv=0, i_UVFLX_coriolis=0, i_UVFLX_pre_grad=0, i_POTT_hor_adv=0,
                                    i_moist_vert_adv=0, i_moist_num_dif=0)),
    # no column-pressure tendency (dyn_continuity.py:211-214).  i_UVFLX_main_switch = 0 / i_POTT_main_switch = 0 cannot be
    # generated: the reference's add_up_tendencies then returns a local it never assigned (dyn_UFLX.py:198, dyn_POTT.py:110)
    'g10x4_termsC': dict(d=10, nz=4, steps=2, snaps=(1, 2), stages=True, turb=False, terms=dict(i_COLP_main_switch=0)),
    'g5x8_preset': dict(preset=True, d=5, nz=8, lat0=-80, lat1=80, sim_n_days=0.72, out_nth_hour=8),
}

NUMBA_SHIM = r'''
import numpy as _np
class _OOB(_np.ndarray):
    """ndarray view with numba's unchecked indexing: an out-of-range scalar read returns
    whatever lies at that flat offset of the (C-contiguous) allocation -- e.g. QV[i,j,nz] is
    QV[i,j+1,0] -- and NaN if the offset leaves the allocation (genuinely undefined)."""
    def __getitem__(self, idx):
        try:
            return _np.ndarray.__getitem__(self, idx)
        except IndexError:
            try:
                base = self.base if self.base is not None else self
                flat = 0
                for n, st in zip(idx, self.strides):
                    flat += int(n) * (st // self.itemsize)
                fb = _np.asarray(base).reshape(-1)
                if 0 <= flat < fb.size and self.flags.c_contiguous:
                    return fb[flat]
            except Exception:
                pass
            return _np.float64('nan')
def _wrap(f):
    def g(*args):
        args = tuple(a.view(_OOB) if type(a) is _np.ndarray else a for a in args)
        try:
            return f(*args)
        except UnboundLocalError:
            return 0.0
    g.__name__ = getattr(f, '__name__', 'g')
    g.__wrapped__ = f
    return g
def njit(*args, **kwargs):
    if args and callable(args[0]):
        return _wrap(args[0])
    return lambda f: _wrap(f)
jit = njit
prange = range
def vectorize(*args, **kwargs):
    return lambda f: _np.vectorize(f)
float32 = 'float32'
float64 = 'float64'
class _Cuda:
    def __getattr__(self, name):
        raise RuntimeError('numba.cuda stand-in: GPU path not available')
cuda = _Cuda()
class config:
    NUMBA_DEFAULT_NUM_THREADS = 1
'''

NETCDF4_SHIM = r'''
from scipy.io import netcdf_file as _ncf
import numpy as _np
class _Var:
    def __init__(self, v):
        self._v = v
    def __getitem__(self, idx):
        return _np.array(self._v.data[idx], dtype=_np.float64)
class Dataset:
    def __init__(self, filename, mode='r', format=None):
        self._f = _ncf(filename, 'r', mmap=False)
    def __getitem__(self, name):
        return _Var(self._f.variables[name])
'''

NAMELIST_SHIM = r'''
import runpy as _runpy
_g = _runpy.run_path(%(ref)r + '/namelist.py')
globals().update({k: v for k, v in _g.items() if not k.startswith('__')})
# ---- run-mode overrides (SURVEY.md 8a "dry-dycore definition") ----
i_comp_mode = 1
i_load_from_restart = 0
i_save_to_restart = 0
i_load_from_IC = 0
i_radiation = %(radiation)d
i_microphysics = 0
i_turbulence = %(turbulence)d
i_surface_scheme = %(radiation)d      # io_read_namelist.py:91-92 insists on it with radiation; the preset case detaches it again
working_precision = %(wp)r
nz = %(nz)d
lat0_deg = %(lat0)r
lat1_deg = %(lat1)r
dlat_deg = %(d)r
dlon_deg = %(d)r
i_sim_n_days = %(sim_n_days)r
i_out_nth_hour = %(out_nth_hour)r
# ---- namelist.py:286-316 table rule, re-applied for the overridden dlat ----
UVFLX_dif_coef = 0
POTT_dif_coef = 1E-5
COLP_dif_coef = 0
moist_dif_coef = POTT_dif_coef
_tab = {10: 1.5, 8: 1.5, 6: 1.8, 5: 2, 4: 2.5, 3: 3.3, 2: 5, 1.5: 7.5}
if dlat_deg in _tab:
    UVFLX_dif_coef = _tab[dlat_deg]
elif dlat_deg <= 1:
    UVFLX_dif_coef = 10
UVFLX_dif_coef *= 2.0
'''


def _install_shims(tmp, d, nz, wp, turbulence=0, radiation=0, lat0=-90, lat1=90, sim_n_days=1, out_nth_hour=1, terms=None):
    os.makedirs(os.path.join(tmp, 'numba'))
    with open(os.path.join(tmp, 'numba', '__init__.py'), 'w') as f:
        f.write(NUMBA_SHIM)
    with open(os.path.join(tmp, 'netCDF4.py'), 'w') as f:
        f.write(NETCDF4_SHIM)
    os.makedirs(os.path.join(tmp, 'bin'))
    open(os.path.join(tmp, 'bin', '__init__.py'), 'w').close()
    with open(os.path.join(tmp, 'bin', 'rad_longwave_cython.py'), 'w') as f:
        for fn in ('calc_planck_intensity_c', 'calc_surface_emission_c',
                   'rad_calc_LW_RTE_matrix_c'):
            f.write('def %s(*a, **k):\n    raise NotImplementedError\n' % fn)
    with open(os.path.join(tmp, 'namelist.py'), 'w') as f:
        f.write(NAMELIST_SHIM % dict(ref=REF, d=d, nz=nz, wp=wp, turbulence=int(turbulence), radiation=int(radiation),
                                     lat0=lat0, lat1=lat1, sim_n_days=sim_n_days, out_nth_hour=out_nth_hour))
        for k, v in (terms or {}).items():       # per-term switches of namelist.py:62-88
            f.write('%s = %d\n' % (k, v))

    # scipy.interpolate.interp2d (removed in SciPy >= 1.14): bilinear, nearest outside.
    import scipy.interpolate as si
    from scipy.interpolate import RegularGridInterpolator

    class interp2d:
        def __init__(self, x, y, z, kind='linear'):
            x = np.asarray(x, dtype=np.float64).ravel()
            y = np.asarray(y, dtype=np.float64).ravel()
            z = np.asarray(z, dtype=np.float64)
            ix, iy = np.argsort(x), np.argsort(y)
            self.x, self.y = x[ix], y[iy]
            self.rgi = RegularGridInterpolator((self.y, self.x), z[np.ix_(iy, ix)],
                                               method='linear')

        def __call__(self, xn, yn):
            xn = np.sort(np.atleast_1d(np.asarray(xn, dtype=np.float64)))
            yn = np.sort(np.atleast_1d(np.asarray(yn, dtype=np.float64)))
            xc = np.clip(xn, self.x[0], self.x[-1])
            yc = np.clip(yn, self.y[0], self.y[-1])
            yy, xx = np.meshgrid(yc, xc, indexing='ij')
            return self.rgi(np.stack([yy.ravel(), xx.ravel()], -1)).reshape(len(yn), len(xn))
    si.interp2d = interp2d


GRID_CONSTS = ['corf', 'corf_is', 'A', 'sigma_vb', 'dsigma', 'dxjs', 'dyis',
               'lat_rad', 'lat_is_rad', 'dlat_rad', 'dlon_rad',
               'POTT_dif_coef', 'UVFLX_dif_coef', 'moist_dif_coef']
PROGNOSTIC = ['COLP', 'UWIND', 'VWIND', 'POTT', 'QV', 'QC']


def run_preset_case(name):
    """solver.py:82-226 on the reference's preset-1 grid with its own Grid / ModelFields / Turbulence /
    Radiation objects; snapshots at the history steps."""
    cfg = CASES[name]
    sys.dont_write_bytecode = True
    tmp = tempfile.mkdtemp(prefix='golden_shims_')
    _install_shims(tmp, cfg['d'], cfg['nz'], 'float64', turbulence=1, radiation=1, lat0=cfg['lat0'], lat1=cfg['lat1'],
                   sim_n_days=cfg['sim_n_days'], out_nth_hour=cfg['out_nth_hour'])
    sys.path.insert(0, REF)
    sys.path.insert(0, tmp)
    os.chdir(REF)
    from datetime import timedelta
    from io_read_namelist import CPU
    from main_grid import Grid
    import main_fields
    import dyn_matsuno
    import rad_longwave
    import rad_main
    from dyn_org_discretizations import DiagnosticsFactory
    from namelist import pair_top
    rad_longwave.run_cython = 0
    # ModelFields.__init__ runs a first radiation call (main_fields.py:94-101) on the surface scheme's
    # SOILTEMP; with the surface scheme off that field is NaN, so the radiation object is attached below
    # (same order: after Turbulence / Microphysics, before the loop) once its inputs are set
    main_fields.i_radiation = 0
    main_fields.i_surface_scheme = 0            # no Surface object: SOILTEMP, albedos and the surface fluxes are fixed inputs
    GR = Grid()
    F = main_fields.ModelFields(GR, False)
    Diagnostics = DiagnosticsFactory(target=CPU)
    h = F.host
    for n in ['SSHFLX', 'SLHFLX', 'SMOMXFLX', 'SMOMYFLX', 'dPOTTdt_RAD']:
        h[n][:] = 0.
    # surface inputs: the surface scheme's initial state (srfc_main.py:44-72), held fixed
    lat = GR.lat_rad[:, :, 0]
    h['SOILTEMP'][:, :, 0] = 295 - np.sin(lat) ** 2 * 30 - h['HSURF'][:, :, 0] * 0.01
    h['SURFALBEDLW'][:] = 0.
    h['SURFALBEDSW'][:, :, 0] = np.where(h['HSURF'][:, :, 0] <= 100, 0.05, 0.2)
    h['SURFALBEDSW'][:, :, 0][h['SOILTEMP'][:, :, 0] <= 273.15] = 0.6
    out = {}
    out['meta/nx'], out['meta/ny'], out['meta/nz'] = np.int64(GR.nx), np.int64(GR.ny), np.int64(GR.nz)
    out['meta/dt'] = np.int64(GR.dt)
    out['meta/pair_top'] = np.float64(pair_top)
    out['meta/dlat_deg'] = np.float64(cfg['d'])
    out['meta/lat0_deg'], out['meta/lat1_deg'] = np.float64(cfg['lat0']), np.float64(cfg['lat1'])
    out['meta/i_out_nth_ts'] = np.int64(GR.i_out_nth_ts)
    out['meta/nts'] = np.float64(GR.nts)
    for n in GRID_CONSTS:
        out['grid/' + n] = np.array(GR.GRF[CPU][n])
    for n in PROGNOSTIC + ['HSURF', 'WWIND', 'POTTVB', 'SSHFLX', 'SLHFLX', 'SMOMXFLX', 'SMOMYFLX',
                           'SOILTEMP', 'SURFALBEDLW', 'SURFALBEDSW']:
        out['ic/' + n] = h[n].copy()
    RAD = rad_main.Radiation(GR)
    F.RAD = RAD
    RAD.calc_radiation(GR, F)                   # main_fields.py:101
    out['meta/rad_nth_ts'] = np.int64(RAD.rad_nth_ts)
    out['ic_rad/dPOTTdt_RAD'] = h['dPOTTdt_RAD'].copy()
    Diagnostics.primary_diag(GR.GRF[CPU], **F.get(Diagnostics.fields_primary_diag, target=CPU))    # solver.py:70-73
    SNAP = ['UWIND', 'VWIND', 'WWIND', 'COLP', 'PHI', 'POTT', 'QV', 'QC', 'SWFLXNET', 'LWFLXNET', 'dPOTTdt_RAD',
            'KMOM', 'KHEAT', 'dVFLXdt_TURB', 'TAIR', 'RHO', 'WIND']
    n_out = 2
    last = n_out * GR.i_out_nth_ts
    track = np.zeros((last, 3))
    import time as _time
    t0 = _time.time()
    while GR.ts < last:
        GR.ts += 1                              # solver.py:86-90
        GR.sim_time_sec = GR.ts * GR.dt
        GR.GMT += timedelta(seconds=GR.dt)
        Diagnostics.secondary_diag(**F.get(Diagnostics.fields_secondary_diag, target=CPU))
        F.TURB.compute_turbulence(GR, **F.get(F.TURB.fields_main, target=CPU))
        RAD.launch_radiation_calc(GR, F)
        dyn_matsuno.step_matsuno(GR, F)
        umax = np.nanmax(np.abs(h['UWIND'][1:-1, 1:-1, :]))
        track[GR.ts - 1] = (umax, np.nanmean(h['COLP'][1:-1, 1:-1, 0]), np.nanmean(h['POTT'][1:-1, 1:-1, :]))
        if GR.ts % 10 == 0 or GR.ts == 1:
            print('[%s] step %d / %d  max|U| = %.6f  mean COLP = %.3f  (%.0f s)' % (
                name, GR.ts, last, umax, track[GR.ts - 1, 1], _time.time() - t0), flush=True)
        if not np.isfinite(umax):
            raise RuntimeError('reference run produced non-finite UWIND')
        if GR.ts % GR.i_out_nth_ts == 0:        # solver.py:152-164
            tag = 'out%04d' % (GR.ts // GR.i_out_nth_ts)
            Diagnostics.secondary_diag(**F.get(Diagnostics.fields_secondary_diag, target=CPU)) if False else None
            for n in SNAP:
                out['%s/%s' % (tag, n)] = h[n].copy()
            out['%s/ts' % tag] = np.int64(GR.ts)
    out['track/umax_colp_pott'] = track
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


def run_case(name):
    cfg = CASES[name]
    if cfg.get('preset'):
        return run_preset_case(name)
    sys.dont_write_bytecode = True
    tmp = tempfile.mkdtemp(prefix='golden_shims_')
    _install_shims(tmp, cfg['d'], cfg['nz'], cfg.get('wp', 'float64'), cfg.get('turbulence', False), terms=cfg.get('terms'))
    sys.path.insert(0, REF)
    sys.path.insert(0, tmp)
    os.chdir(REF)            # the reference opens data/... relative to cwd

    from io_read_namelist import CPU
    from main_grid import Grid
    from main_fields import ModelFields
    import dyn_tendencies
    import dyn_matsuno
    from dyn_org_discretizations import DiagnosticsFactory
    from namelist import pair_top

    GR = Grid()
    F = ModelFields(GR, False)
    Diagnostics = DiagnosticsFactory(target=CPU)
    T = dyn_tendencies.Tendencies
    P = dyn_matsuno.Prognostics
    D = dyn_matsuno.Diagnostics

    turbulence = cfg.get('turbulence', False)
    hot = []
    extra = [F.TURB.fields_main] if turbulence else []
    for lst in [T.fields_continuity, T.fields_momentum, T.fields_temperature,
                T.fields_moisture, D.fields_primary_diag, D.fields_secondary_diag,
                P.fields_prognostic] + extra:
        for n in lst:
            if n not in hot:
                hot.append(n)

    rng = np.random.default_rng(11)
    turbulence = cfg.get('turbulence', False)
    # physics-produced inputs: 0 in the dry definition, random in the turb variant; with the
    # turbulence scheme on, KMOM/KHEAT stay as allocated (NaN) until compute_turbulence fills
    # levels 1..nz-1 (the end levels are never consumed)
    for n in ['KMOM', 'KHEAT', 'SSHFLX', 'SLHFLX', 'SMOMXFLX', 'SMOMYFLX', 'dPOTTdt_RAD']:
        if turbulence and n in ('KMOM', 'KHEAT'):
            continue
        F.host[n][:] = 0.
    if cfg['turb']:
        F.host['KMOM'][:] = rng.uniform(0.5, 20., F.host['KMOM'].shape)
        F.host['KHEAT'][:] = rng.uniform(0.5, 20., F.host['KHEAT'].shape)
        F.host['SSHFLX'][:] = rng.uniform(-50., 150., F.host['SSHFLX'].shape)
        F.host['SLHFLX'][:] = rng.uniform(-20., 200., F.host['SLHFLX'].shape)
        F.host['SMOMXFLX'][:] = rng.uniform(-0.2, 0.2, F.host['SMOMXFLX'].shape)
        F.host['SMOMYFLX'][:] = rng.uniform(-0.2, 0.2, F.host['SMOMYFLX'].shape)

    out = {}
    out['meta/nx'] = np.int64(GR.nx)
    out['meta/ny'] = np.int64(GR.ny)
    out['meta/nz'] = np.int64(GR.nz)
    out['meta/nb'] = np.int64(GR.nb)
    out['meta/dt'] = np.int64(GR.dt)
    out['meta/pair_top'] = np.float64(pair_top)
    out['meta/dlat_deg'] = np.float64(cfg['d'])
    if cfg.get('terms'):        # the namelist entries this case sets to 0
        out['meta/terms_off'] = np.array(sorted(k for k, v in cfg['terms'].items() if not v))
    for n in GRID_CONSTS:
        out['grid/' + n] = np.array(GR.GRF[CPU][n])
    out['ic/HSURF'] = F.host['HSURF'].copy()
    for n in PROGNOSTIC + ['KMOM', 'KHEAT', 'SSHFLX', 'SLHFLX', 'SMOMXFLX', 'SMOMYFLX',
                           'dPOTTdt_RAD', 'WWIND', 'POTTVB']:
        out['ic/' + n] = F.host[n].copy()

    # solver.py:70-73
    Diagnostics.primary_diag(GR.GRF[CPU], **F.get(Diagnostics.fields_primary_diag, target=CPU))

    # ---------------- stage capture (first step only) ----------------
    state = {'last': {}, 'seq': []}

    def capture(tag, full=False):
        for n in hot:
            a = F.host[n]
            prev = state['last'].get(n)
            if full or prev is None or not np.array_equal(a, prev, equal_nan=True):
                out['%s/%s' % (tag, n)] = a.copy()
                state['last'][n] = a.copy()
        state['seq'].append(tag)

    def hook(obj, meth, label):
        orig = getattr(obj, meth)

        def wrapped(*a, **k):
            r = orig(*a, **k)
            if state.get('on'):
                state['n'] = state.get('n', 0) + 1
                capture('st%02d_%s' % (state['n'], label))
            return r
        setattr(obj, meth, wrapped)

    if cfg['stages']:
        hook(T, 'continuity', 'continuity')
        hook(T, 'momentum', 'momentum')
        hook(T, 'temperature', 'temperature')
        hook(T, 'moisture', 'moisture')
        hook(P, 'euler_forward', 'euler_forward')
        hook(D, 'primary_diag', 'primary_diag')

    if turbulence:
        hook(Diagnostics, 'secondary_diag', 'secondary_diag')
        hook(F.TURB, 'compute_turbulence', 'turbulence')
    for ts in range(1, cfg['steps'] + 1):
        if turbulence:      # solver.py:99-113 with i_turbulence = 1
            if ts == 1 and cfg['stages']:
                capture('st00_pre', full=True)
                state['on'] = True
            Diagnostics.secondary_diag(**F.get(Diagnostics.fields_secondary_diag, target=CPU))
            F.TURB.compute_turbulence(GR, **F.get(F.TURB.fields_main, target=CPU))
        else:
            Diagnostics.secondary_diag(**F.get(Diagnostics.fields_secondary_diag, target=CPU))
        if ts == 1 and cfg['stages'] and not turbulence:
            capture('st00_pre', full=True)
            state['on'] = True
        dyn_matsuno.step_matsuno(GR, F)
        state['on'] = False
        if ts in cfg['snaps']:
            for n in PROGNOSTIC:
                out['step%03d/%s' % (ts, n)] = F.host[n].copy()
        umax = np.nanmax(np.abs(F.host['UWIND'][1:-1, 1:-1, :]))
        print('[%s] step %d  max|U| = %.6f  mean COLP = %.3f' % (
            name, ts, umax, np.nanmean(F.host['COLP'][1:-1, 1:-1, 0])), flush=True)
        if not np.isfinite(umax):
            raise RuntimeError('reference run produced non-finite UWIND')
    if cfg['stages']:
        out['meta/stage_seq'] = np.array(state['seq'])

    if turbulence:
        # kernel-level vector for compute_turbulence_cpu (turb_compute.py:191-208): in the model
        # state above every K sits on the lower clamp; random near-surface columns exercise the
        # unclamped formula, both clamps and the equal-wind guard
        import turb_compute
        r2 = np.random.default_rng(23)
        shp, shps = F.host['POTT'].shape, F.host['PHIVB'].shape
        kin = {}
        kin['HSURF'] = r2.uniform(0., 2000., F.host['HSURF'].shape)
        hgt = np.cumsum(r2.uniform(0.02, 1.5, shps)[:, :, ::-1], axis=2)[:, :, ::-1]     # decreasing with k
        kin['PHIVB'] = (kin['HSURF'] + hgt - hgt[:, :, -1:]) * 9.81
        kin['PHI'] = 0.5 * (kin['PHIVB'][:, :, :-1] + kin['PHIVB'][:, :, 1:])
        kin['QV'] = r2.uniform(0., 0.02, shp)
        kin['QV'][r2.uniform(size=shp) < 0.1] = 0.                                        # below the 1e-7 clamp
        amp = 10. ** r2.uniform(-3., 0.5, shp[:2] + (1,))                                 # per-column shear scale
        kin['WINDX'] = 5. + amp * r2.uniform(-1., 1., shp)
        kin['WINDY'] = -3. + amp * r2.uniform(-1., 1., shp)
        same = r2.uniform(size=shp[:2]) < 0.1
        kin['WINDX'][same] = kin['WINDX'][same][:, :1]                                    # equal winds in a column
        kin['POTT'] = 300. - 1e-4 * np.arange(shp[2])[None, None, :] + r2.uniform(-2e-4, 2e-4, shp)
        kin['POTTVB'] = 300. + r2.uniform(-0.5, 0.5, shps)
        KM = np.full(shps, np.nan)
        KH = np.full(shps, np.nan)
        turb_compute.compute_turbulence_cpu(KM, KH, kin['PHIVB'].copy(), kin['HSURF'].copy(), kin['PHI'].copy(),
                                            kin['QV'].copy(), kin['WINDX'].copy(), kin['WINDY'].copy(),
                                            kin['POTTVB'].copy(), kin['POTT'].copy())
        for n, a in kin.items():
            out['kturb_in/' + n] = a
        out['kturb_out/KMOM'] = KM
        out['kturb_out/KHEAT'] = KH
        inner = KM[:, :, 1:-1]
        print('[%s] kernel-level turbulence vector: %.0f%% at min, %.0f%% at max, %.0f%% in between' % (
            name, 100 * np.mean(inner == 1e-6), 100 * np.mean(inner == 0.01),
            100 * np.mean((inner > 1e-6) & (inner < 0.01))))

    if cfg.get('radiation'):
        # rad_main.py:121-205 through the reference's own class; the cython helpers of
        # rad_longwave are absent, so its pure-numpy branch (run_cython = 0) computes the Planck
        # sums and the matrix; scipy.linalg.solve_banded (LAPACK gbsv / gtsv) solves the columns
        import datetime
        import rad_longwave
        import rad_main
        rad_longwave.run_cython = 0
        Diagnostics.secondary_diag(**F.get(Diagnostics.fields_secondary_diag, target=CPU))
        RAD = rad_main.Radiation(GR)
        GR.GMT = datetime.datetime(*cfg['radiation'])
        r3 = np.random.default_rng(31)
        fin = {n: F.host[n] for n in RAD.fields_radiation}
        lat = GR.lat_rad[:, :, 0]
        fin['SOILTEMP'][:] = np.nan
        fin['SOILTEMP'][:, :, 0] = 255. + 45. * np.cos(lat) + r3.uniform(-3., 3., lat.shape)
        fin['SURFALBEDLW'][:, :, 0] = r3.uniform(0., 0.3, lat.shape)
        fin['SURFALBEDSW'][:, :, 0] = r3.uniform(0.05, 0.8, lat.shape)
        fin['QC'][:] = r3.uniform(0., 0.004, fin['QC'].shape) * (r3.uniform(size=fin['QC'].shape) < 0.5)   # half clear, some above the 0.002 cap
        outs = ['LWFLXDO', 'LWFLXUP', 'SWDIFFLXDO', 'SWDIRFLXDO', 'SWFLXUP', 'SWFLXDO', 'LWFLXNET', 'SWFLXNET',
                'SWFLXDIV', 'LWFLXDIV', 'TOTFLXDIV', 'dPOTTdt_RAD']
        for n in outs:
            fin[n][:] = np.nan
        for n in ('PHIVB', 'TAIR', 'RHO', 'SOILTEMP', 'SURFALBEDLW', 'SURFALBEDSW', 'QC'):
            out['rad_in/' + n] = fin[n].copy()
        RAD.simple_radiation(GR, **fin)
        out['rad_in/SOLZEN'] = fin['SOLZEN'].copy()            # written in place by rad_solar_zenith_angle
        out['rad_in/MYSUN'] = np.cos(fin['SOLZEN'])            # locals of simple_radiation (rad_main.py:133-137)
        out['rad_in/MYSUN'][GR.ii, GR.jj, 0][out['rad_in/MYSUN'][GR.ii, GR.jj, 0] < 0] = 0   # a no-op on a copy, as in the reference
        out['rad_in/SWINTOA'] = RAD.solar_constant * np.cos(fin['SOLZEN'])
        out['rad_meta/solar_constant'] = np.float64(RAD.solar_constant)
        out['rad_meta/gmt'] = np.array(cfg['radiation'], dtype=np.int64)
        out['rad_meta/planck_lambdas_center'] = np.array(RAD.planck_lambdas_center, dtype=np.float64)
        out['rad_meta/planck_dlambdas'] = np.array(RAD.planck_dlambdas, dtype=np.float64)
        out['rad_meta/lon_rad'] = np.array(GR.lon_rad)
        for n in outs:
            out['rad_out/' + n] = fin[n].copy()
        sun = out['rad_in/MYSUN'][1:-1, 1:-1, 0]
        print('[%s] radiation vector: %.0f%% of the columns sunlit; dPOTTdt_RAD in [%.3e, %.3e] K/s' % (
            name, 100 * np.mean(sun > 0), np.nanmin(fin['dPOTTdt_RAD'][1:-1, 1:-1]), np.nanmax(fin['dPOTTdt_RAD'][1:-1, 1:-1])))

    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name + '.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--case', default=None)
    args = ap.parse_args()
    if args.case:
        run_case(args.case)
        return
    for name in CASES:
        subprocess.run([sys.executable, os.path.abspath(__file__), '--case', name], check=True)


if __name__ == '__main__':
    main()
