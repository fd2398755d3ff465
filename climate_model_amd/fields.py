"""Model-field registry and host/device field container for the dycore hot path.

Mirrors the part of the reference's `main_fields.py` the Matsuno step touches:
the field table (main_fields.py:234-475: staggering and vertical extent of each
array), NaN-initialised host arrays in the reference's C-order [i, j, k] layout
(main_fields.py:477-485), and the get/set/copy_* accessors
(main_fields.py:184-215).  The device side holds `DeviceField` handles naming
buffers that live inside the HIP dycore handle (i-fastest [k][j][i] layout, see
DESIGN.md); the layout adaptor runs on upload/download.
"""
from collections import OrderedDict

import numpy as np

CPU = 'CPU'
HIP = 'HIP'

# name -> (stgx, stgy, vertical extent: 'nz' | 'nzs' | 1)
FIELD_TABLE = OrderedDict([
    # pressure
    ('COLP',          (0, 0, 1)),
    ('COLP_OLD',      (0, 0, 1)),
    ('COLP_NEW',      (0, 0, 1)),
    ('dCOLPdt',       (0, 0, 1)),
    ('HSURF',         (0, 0, 1)),
    # fluxes
    ('UFLX',          (1, 0, 'nz')),
    ('dUFLXdt',       (1, 0, 'nz')),
    ('VFLX',          (0, 1, 'nz')),
    ('dVFLXdt',       (0, 1, 'nz')),
    ('FLXDIV',        (0, 0, 'nz')),
    ('BFLX',          (0, 0, 'nz')),
    ('CFLX',          (1, 1, 'nz')),
    ('DFLX',          (0, 1, 'nz')),
    ('EFLX',          (0, 1, 'nz')),
    ('RFLX',          (0, 0, 'nz')),
    ('QFLX',          (1, 1, 'nz')),
    ('SFLX',          (1, 0, 'nz')),
    ('TFLX',          (1, 0, 'nz')),
    # velocities
    ('UWIND',         (1, 0, 'nz')),
    ('UWIND_OLD',     (1, 0, 'nz')),
    ('VWIND',         (0, 1, 'nz')),
    ('VWIND_OLD',     (0, 1, 'nz')),
    ('WWIND',         (0, 0, 'nzs')),
    ('WWIND_UWIND',   (1, 0, 'nzs')),
    ('WWIND_VWIND',   (0, 1, 'nzs')),
    ('KMOM_dUWINDdz', (1, 0, 'nzs')),
    ('KMOM_dVWINDdz', (0, 1, 'nzs')),
    # temperature
    ('POTT',          (0, 0, 'nz')),
    ('POTT_OLD',      (0, 0, 'nz')),
    ('dPOTTdt',       (0, 0, 'nz')),
    ('POTTVB',        (0, 0, 'nzs')),
    # predictor state of the fused library step (scratch)
    ('UWIND_EST',     (1, 0, 'nz')),
    ('VWIND_EST',     (0, 1, 'nz')),
    ('POTT_EST',      (0, 0, 'nz')),
    # raw turbulent tendencies of the fused turbulence-coupled step (library scratch; device only with i_turb)
    ('TURB_U',        (1, 0, 'nz')),
    ('TURB_V',        (0, 1, 'nz')),
    ('TURB_T',        (0, 0, 'nz')),
    ('TURB_QV',       (0, 0, 'nz')),
    ('TURB_QC',       (0, 0, 'nz')),
    # column pressure at the u- / v-points (2-D factors of UFLX, VFLX; library scratch)
    ('UCOL',          (1, 0, 1)),
    ('VCOL',          (0, 1, 1)),
    # primary diagnostics
    ('PVTF',          (0, 0, 'nz')),
    ('PVTFVB',        (0, 0, 'nzs')),
    ('PHI',           (0, 0, 'nz')),
    ('PHIVB',         (0, 0, 'nzs')),
    # secondary diagnostics / physics inputs the dycore consumes
    ('RHO',           (0, 0, 'nz')),
    ('RHOVB',         (0, 0, 'nzs')),
    ('KMOM',          (0, 0, 'nzs')),
    ('KHEAT',         (0, 0, 'nzs')),
    ('SSHFLX',        (0, 0, 1)),
    ('SMOMXFLX',      (0, 0, 1)),
    ('SMOMYFLX',      (0, 0, 1)),
    ('dPOTTdt_RAD',   (0, 0, 'nz')),
    ('dUFLXdt_TURB',  (1, 0, 'nz')),
    ('dVFLXdt_TURB',  (0, 1, 'nz')),
    ('dPOTTdt_TURB',  (0, 0, 'nz')),
    # moisture (dyn_moist.py)
    ('QV',            (0, 0, 'nz')),
    ('QV_OLD',        (0, 0, 'nz')),
    ('QC',            (0, 0, 'nz')),
    ('QC_OLD',        (0, 0, 'nz')),
    ('dQVdt',         (0, 0, 'nz')),
    ('dQCdt',         (0, 0, 'nz')),
    ('dQVdt_TURB',    (0, 0, 'nz')),
    ('SLHFLX',        (0, 0, 1)),
    ('QV_EST',        (0, 0, 'nz')),
    ('QC_EST',        (0, 0, 'nz')),
    # full secondary diagnostics (dyn_diagnostics.py:199-222)
    ('TAIR',          (0, 0, 'nz')),
    ('TAIRVB',        (0, 0, 'nzs')),
    ('PAIR',          (0, 0, 'nz')),
    ('PAIRVB',        (0, 0, 'nzs')),
    ('WINDX',         (0, 0, 'nz')),
    ('WINDY',         (0, 0, 'nz')),
    ('WIND',          (0, 0, 'nz')),
    # radiation (rad_main.py:33-38); on the device only with i_radiation.
    # SOILTEMP: top soil level (the reference's array has nzsoil levels, radiation reads level 0)
    ('SOILTEMP',      (0, 0, 1)),
    ('SURFALBEDLW',   (0, 0, 1)),
    ('SURFALBEDSW',   (0, 0, 1)),
    ('SOLZEN',        (0, 0, 1)),
    ('MYSUN',         (0, 0, 1)),
    ('SWINTOA',       (0, 0, 1)),
    ('LWFLXDO',       (0, 0, 'nzs')),
    ('LWFLXUP',       (0, 0, 'nzs')),
    ('SWDIFFLXDO',    (0, 0, 'nzs')),
    ('SWDIRFLXDO',    (0, 0, 'nzs')),
    ('SWFLXUP',       (0, 0, 'nzs')),
    ('SWFLXDO',       (0, 0, 'nzs')),
    ('LWFLXNET',      (0, 0, 'nzs')),
    ('SWFLXNET',      (0, 0, 'nzs')),
    ('SWFLXDIV',      (0, 0, 'nz')),
    ('LWFLXDIV',      (0, 0, 'nz')),
    ('TOTFLXDIV',     (0, 0, 'nz')),
])
RADIATION_FIELDS = ('SOILTEMP', 'SURFALBEDLW', 'SURFALBEDSW', 'SOLZEN', 'MYSUN', 'SWINTOA', 'LWFLXDO', 'LWFLXUP',
                    'SWDIFFLXDO', 'SWDIRFLXDO', 'SWFLXUP', 'SWFLXDO', 'LWFLXNET', 'SWFLXNET', 'SWFLXDIV', 'LWFLXDIV',
                    'TOTFLXDIV')

# fields the reference passes through a factory signature that the dycore does not own
PASS_THROUGH_FIELDS = ()


def field_shape(name, nx, ny, nz, nb=1):
    """Host shape (reference layout) of a field: main_fields.py:477-485."""
    stgx, stgy, nk = FIELD_TABLE[name]
    nk = {'nz': nz, 'nzs': nz + 1}.get(nk, nk)
    return (nx + 2 * nb + stgx, ny + 2 * nb + stgy, nk)


class _LazyHostFields(dict):
    """name -> NaN-filled host array (main_fields.py:477-485), allocated on first access:
    at 0.25 deg x 64 levels the full set is 24 GB, of which a driver touches a handful."""

    def __init__(self, nx, ny, nz, dtype):
        super().__init__()
        self._ext = (nx, ny, nz)
        self._dtype = dtype

    def __missing__(self, name):
        if name not in FIELD_TABLE:
            raise KeyError(name)
        a = np.full(field_shape(name, *self._ext), np.nan, dtype=self._dtype)
        self[name] = a
        return a

    def get(self, name, default=None):
        return self[name] if name in FIELD_TABLE else default

    def keys(self):
        return FIELD_TABLE.keys()


def allocate_fields(nx, ny, nz, dtype=np.float64):
    """Host arrays for every hot-path field, NaN-initialised like the reference's."""
    return _LazyHostFields(nx, ny, nz, dtype)


class DeviceField:
    """Names one field buffer that lives inside a HIP dycore handle."""
    __slots__ = ('name', 'core')

    def __init__(self, name, core):
        self.name = name
        self.core = core

    def __repr__(self):
        return 'DeviceField(%s)' % self.name


class ModelFields:
    """Host dict + device mirror; same accessors as main_fields.py:184-215."""

    ALL_FIELDS = 'all_fields'
    PROGNOSTIC_FIELDS = 'prognostic_fields'
    PRINT_DIAG_FIELDS = 'print_diag_fields'

    def __init__(self, GR, core=None, dtype=None):
        if dtype is None:       # the working precision of the handle (namelist.py:225-226), float64 without one
            dtype = getattr(core, 'dtype', None) or getattr(getattr(GR, 'nl', None), 'wp', None) or np.float64
        self.host = allocate_fields(GR.nx, GR.ny, GR.nz, dtype=dtype)
        self.device = {}
        self.core = core
        self.field_groups = {
            self.ALL_FIELDS: list(self.host.keys()),
            self.PROGNOSTIC_FIELDS: ['COLP', 'UWIND', 'VWIND', 'POTT'],
            self.PRINT_DIAG_FIELDS: ['COLP', 'WIND', 'POTT', 'UWIND'],
        }
        # non-NaN values the kernels rely on (io_initial_conditions.py:45-46)
        self.host['POTTVB'][:] = 0.
        self.host['WWIND'][:] = 0.
        # physics-produced inputs: zero in the dry dycore
        for n in ('KMOM', 'KHEAT', 'SSHFLX', 'SLHFLX', 'SMOMXFLX', 'SMOMYFLX', 'dPOTTdt_RAD'):
            self.host[n][:] = 0.
        if core is not None:
            self.attach(core)

    def attach(self, core):
        self.core = core
        self.device = {name: DeviceField(name, core) for name in FIELD_TABLE}
        # main_fields.py:150-182: ALL_FIELDS = every field -- of those this handle holds (the radiation fields
        # exist only with i_radiation, the TURB_* scratch fields only with i_turb)
        has = getattr(core, 'has_field', None)
        if has is not None:
            self.field_groups[self.ALL_FIELDS] = [n for n in FIELD_TABLE if has(n)]
        for name in PASS_THROUGH_FIELDS:
            self.device[name] = None

    def get(self, field_names, target=CPU):
        src = self.host if target == CPU else self.device
        return {name: src.get(name) for name in field_names}

    def set(self, field_dict, target=CPU):
        dst = self.host if target == CPU else self.device
        for name, field in field_dict.items():
            dst[name] = field

    def copy_host_to_device(self, GR=None, field_group=ALL_FIELDS):
        """Upload a field group.  Device buffers are NaN-filled at creation like the host
        arrays, so fields whose host copy was never touched need no transfer."""
        names = self.field_groups[field_group] if isinstance(field_group, str) else list(field_group)
        for name in names:
            if field_group == self.ALL_FIELDS and not dict.__contains__(self.host, name):
                continue
            self.core.upload(name, self.host[name])

    def copy_device_to_host(self, GR=None, field_group=ALL_FIELDS):
        names = self.field_groups[field_group] if isinstance(field_group, str) else list(field_group)
        for name in names:
            self.core.download(name, self.host[name])
