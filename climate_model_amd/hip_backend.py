"""ctypes binding of libdycore_hip.so (C ABI: include/dycore.h).

This is the only compute backend of the package.  There is no CPU fallback: if the
shared library is missing or no MI355X is visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libdycore_hip.so')
LIB_PATH_F32 = os.path.join(_HERE, 'csrc', 'libdycore_hip_f32.so')     # working_precision = 'float32' (namelist.py:225-226)
ABI_VERSION = 1


class DycoreError(RuntimeError):
    pass


class DycoreConfig(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('nx', C.c_int32), ('ny', C.c_int32), ('nz', C.c_int32),
                ('ny_global', C.c_int32), ('j_offset', C.c_int32), ('device', C.c_int32),
                ('i_turb', C.c_int32), ('i_radiation', C.c_int32), ('i_moist', C.c_int32), ('i_moist_clip', C.c_int32),
                ('term_off', C.c_int32), ('reserved', C.c_int32 * 4),
                ('pair_top', C.c_double), ('stream', C.c_void_p)]


# every symbol include/dycore.h declares (tests check the library exports all of them)
ABI_SYMBOLS = [
    'dycore_create', 'dycore_destroy', 'dycore_last_error', 'dycore_abi_version', 'dycore_real_bytes', 'dycore_sync',
    'dycore_set_grid', 'dycore_has_field', 'dycore_field_shape', 'dycore_upload', 'dycore_download', 'dycore_field_ptr',
    'dycore_copy_field', 'dycore_exchange_bc',
    'dycore_continuity', 'dycore_momentum', 'dycore_momentum_prepare', 'dycore_uflx_tendency',
    'dycore_vflx_tendency', 'dycore_pott_tendency', 'dycore_moist_tendency', 'dycore_euler_forward',
    'dycore_primary_diag',
    'dycore_secondary_diag_rho', 'dycore_secondary_diag', 'dycore_compute_turbulence', 'dycore_radiation',
    'dycore_step_matsuno', 'dycore_time_steps',
    'dycore_step_begin', 'dycore_substep_part', 'dycore_use_stream', 'dycore_get_stream', 'dycore_overlap_parts', 'dycore_flux_from_winds', 'dycore_rows_size', 'dycore_pack_rows',
    'dycore_unpack_rows', 'dycore_substep_field', 'dycore_band_row_fields',
    'dycore_profile', 'dycore_n_kernels', 'dycore_kernel_stats',
    'dycore_bands_unique_id', 'dycore_bands_create_rccl', 'dycore_bands_create_local', 'dycore_bands_connect',
    'dycore_bands_step_matsuno', 'dycore_bands_destroy',
]

_lib = None
_lib_f32 = None
_runtime_preloaded = None


def _one_hip_runtime():
    """One HIP / HSA runtime per process, brought up in the order PyTorch-ROCm needs.
    libdycore_hip.so asks for `libamdhip64.so.7` (/opt/rocm/lib); a PyTorch-ROCm wheel ships its own copy and
    asks for it as `libamdhip64.so` (same for libhsa-runtime64).  The loader matches an already loaded library
    by SONAME or by file, so: torch first -> this library binds to torch's copy, one runtime; this library first
    -> a later `import torch` maps its own copy NEXT TO the system one, two HSA runtimes share the process and
    torch.cuda comes up with "No HIP GPUs are available" -- the failure that used to depend on which test module
    pytest happened to import first (tools/probe_runtime_order.py shows both cases).  Loading torch's copy by
    hand before this library is no way out either: torch's libraries then register their code objects with an
    already initialised runtime, which loads them eagerly (minutes).  So when torch is installed it is imported
    before libdycore_hip.so is mapped; a host without torch (plain C: examples/c_client.c) runs on the system
    runtime.  DYCORE_NO_TORCH_PRELOAD=1 switches this off (probe only)."""
    global _runtime_preloaded
    if _runtime_preloaded is not None or os.environ.get('DYCORE_NO_TORCH_PRELOAD'):
        return
    _runtime_preloaded = False
    import sys
    if 'torch' in sys.modules:
        return
    try:
        import importlib.util
        if importlib.util.find_spec('torch') is None:
            return
        import torch  # noqa: F401  (its bundled HIP runtime becomes the process's one)
        _runtime_preloaded = True
    except Exception:
        pass


def load_library(path=None, dtype=np.float64):
    """Load libdycore_hip.so (or, for dtype float32, libdycore_hip_f32.so) and declare its prototypes.
    Raises if it is absent."""
    global _lib, _lib_f32
    f32 = np.dtype(dtype) == np.float32
    if path is None:
        if f32 and _lib_f32 is not None:
            return _lib_f32
        if not f32 and _lib is not None:
            return _lib
    cached = path is None
    path = path or ((os.environ.get('DYCORE_LIB_F32') or LIB_PATH_F32) if f32 else
                    (os.environ.get('DYCORE_LIB') or LIB_PATH))   # DYCORE_LIB: A/B builds of the same ABI
    if not os.path.exists(path):
        raise DycoreError('HIP extension not built: %s is missing (run __graft_entry__.build() '
                          'or `make -C climate_model_amd/csrc`); there is no CPU fallback' % path)
    _one_hip_runtime()
    lib = C.CDLL(path)
    hp = C.c_void_p
    dp = C.POINTER(C.c_float if f32 else C.c_double)        # dycore_real_t of that build
    lib.dycore_last_error.restype = C.c_char_p
    lib.dycore_create.argtypes = [C.POINTER(DycoreConfig), C.POINTER(hp)]
    lib.dycore_destroy.argtypes = [hp]
    lib.dycore_sync.argtypes = [hp]
    lib.dycore_set_grid.argtypes = [hp, C.c_char_p, dp, C.c_int32, C.c_int32, C.c_int32]
    lib.dycore_field_shape.argtypes = [hp, C.c_char_p, C.POINTER(C.c_int32 * 3)]
    if hasattr(lib, 'dycore_has_field'):
        lib.dycore_has_field.argtypes = [hp, C.c_char_p]
    lib.dycore_upload.argtypes = [hp, C.c_char_p, dp]
    lib.dycore_download.argtypes = [hp, C.c_char_p, dp]
    lib.dycore_field_ptr.argtypes = [hp, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.dycore_copy_field.argtypes = [hp, C.c_char_p, C.c_char_p]
    lib.dycore_exchange_bc.argtypes = [hp, C.c_char_p]
    for n in ('dycore_continuity', 'dycore_euler_forward'):
        getattr(lib, n).argtypes = [hp, C.c_double]
    for n in ('dycore_momentum', 'dycore_momentum_prepare', 'dycore_uflx_tendency',
              'dycore_vflx_tendency', 'dycore_pott_tendency', 'dycore_moist_tendency', 'dycore_primary_diag',
              'dycore_secondary_diag_rho', 'dycore_secondary_diag', 'dycore_compute_turbulence'):
        getattr(lib, n).argtypes = [hp]
    lib.dycore_radiation.argtypes = [hp, C.c_double, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.dycore_step_matsuno.argtypes = [hp, C.c_double, C.c_int32, C.c_int32]
    lib.dycore_time_steps.argtypes = [hp, C.c_double, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    lib.dycore_step_begin.argtypes = [hp, C.c_int32]
    lib.dycore_substep_part.argtypes = [hp, C.c_double, C.c_int32, C.c_int32]
    lib.dycore_use_stream.argtypes = [hp, C.c_void_p]
    lib.dycore_overlap_parts.argtypes = [hp]
    lib.dycore_flux_from_winds.argtypes = [hp]
    lib.dycore_get_stream.argtypes = [hp]
    lib.dycore_get_stream.restype = C.c_void_p
    lib.dycore_substep_field.argtypes = [hp, C.c_int32, C.c_char_p]
    lib.dycore_substep_field.restype = C.c_char_p
    if hasattr(lib, 'dycore_band_row_fields'):
        lib.dycore_band_row_fields.argtypes = [hp, C.c_char_p, C.c_int32]
    cpp = C.POINTER(C.c_char_p)
    lib.dycore_rows_size.argtypes = [hp, C.c_int32, cpp, C.POINTER(C.c_int64)]
    lib.dycore_pack_rows.argtypes = [hp, C.c_int32, cpp, C.POINTER(C.c_int32), C.c_void_p]
    lib.dycore_unpack_rows.argtypes = [hp, C.c_int32, cpp, C.POINTER(C.c_int32), C.c_void_p]
    if hasattr(lib, 'dycore_bands_step_matsuno'):       # absent from older builds loaded through DYCORE_LIB for A/B timing
        lib.dycore_bands_unique_id.argtypes = [C.c_void_p, C.c_int32]
        lib.dycore_bands_create_rccl.argtypes = [hp, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(hp)]
        lib.dycore_bands_create_local.argtypes = [C.POINTER(hp), C.c_int32, C.POINTER(hp)]
        lib.dycore_bands_connect.argtypes = [hp]
        lib.dycore_bands_step_matsuno.argtypes = [hp, C.c_double, C.c_int32, C.c_int32]
        lib.dycore_bands_destroy.argtypes = [hp]
    lib.dycore_profile.argtypes = [hp, C.c_int32]
    lib.dycore_kernel_stats.argtypes = [hp, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                        C.POINTER(C.c_double)]
    if lib.dycore_abi_version() != ABI_VERSION:
        raise DycoreError('libdycore_hip.so ABI %d != binding ABI %d' % (lib.dycore_abi_version(), ABI_VERSION))
    if hasattr(lib, 'dycore_real_bytes') and lib.dycore_real_bytes() != (4 if f32 else 8):
        raise DycoreError('%s holds %d-byte fields, asked for %s' % (path, lib.dycore_real_bytes(), np.dtype(dtype).name))
    lib._real = C.c_float if f32 else C.c_double
    if cached:
        if f32:
            _lib_f32 = lib
        else:
            _lib = lib
    return lib


GRID_ARRAYS = ['A', 'dxjs', 'dyis', 'corf', 'corf_is', 'lat_rad', 'lat_is_rad', 'dlon_rad',
               'dlat_rad', 'sigma_vb', 'dsigma', 'UVFLX_dif_coef', 'POTT_dif_coef']


class HipDycore:
    """One dycore handle = one GPU (or one latitude band of a decomposed grid)."""

    def __init__(self, nx, ny, nz, pair_top=10000., i_turb=0, i_radiation=0, device=0,
                 ny_global=None, j_offset=0, stream=None, i_moist=0, i_moist_clip=0, dtype=np.float64, term_off=0):
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("working precision is 'float64' or 'float32' (io_read_namelist.py:47-53)")
        self.lib = load_library(dtype=self.dtype)
        self._rp = C.POINTER(self.lib._real)
        self.nx, self.ny, self.nz = int(nx), int(ny), int(nz)
        cfg = DycoreConfig()
        cfg.abi_version = ABI_VERSION
        cfg.nx, cfg.ny, cfg.nz = self.nx, self.ny, self.nz
        cfg.ny_global = int(ny_global if ny_global is not None else ny)
        cfg.j_offset = int(j_offset)
        cfg.device = int(device)
        cfg.i_turb = int(bool(i_turb))
        cfg.i_radiation = int(bool(i_radiation))
        cfg.i_moist = int(bool(i_moist))
        cfg.i_moist_clip = int(bool(i_moist_clip))
        cfg.term_off = int(term_off)        # DYCORE_TERM_* bits (namelist.TERM_BITS): tendency terms switched off
        cfg.pair_top = float(pair_top)
        cfg.stream = C.c_void_p(stream) if stream else None
        self._h = C.c_void_p()
        self._check(self.lib.dycore_create(C.byref(cfg), C.byref(self._h)))
        self.i_turb = bool(i_turb)
        self.i_moist = bool(i_moist)
        self.i_radiation = bool(i_radiation)
        self.term_off = int(term_off)

    @classmethod
    def from_namelist(cls, GR, **kw):
        """A handle configured the way the reference's kernels are by `namelist.py`: extents from the grid, the coupling
        and per-term switches (namelist.py:62-88), pair_top and the working precision from GR.nl."""
        nl = GR.nl
        args = dict(pair_top=nl.pair_top, i_turb=nl.i_turb, i_radiation=nl.i_POTT_radiation, i_moist=nl.i_moist_main_switch,
                    i_moist_clip=nl.i_moist_clip, term_off=nl.term_off, dtype=nl.wp)
        ny = kw.pop('ny', GR.ny)            # a latitude band passes its own height (+ ny_global, j_offset)
        args.update(kw)
        return cls(GR.nx, ny, GR.nz, **args)

    # ---- plumbing ----
    def _check(self, rc):
        if rc != 0:
            raise DycoreError('libdycore_hip: %s (code %d)' % (self.lib.dycore_last_error().decode(), rc))

    def close(self):
        if getattr(self, '_h', None) and self._h.value:
            self.lib.dycore_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._check(self.lib.dycore_sync(self._h))

    # ---- grid constants / fields ----
    def set_grid(self, GRF):
        names = GRID_ARRAYS + (['moist_dif_coef'] if self.i_moist else [])
        for name in names:
            a = np.ascontiguousarray(np.asarray(GRF[name], dtype=self.dtype))
            shp = a.shape if a.ndim == 3 else (a.size, 1, 1)
            self._check(self.lib.dycore_set_grid(self._h, name.encode(), a.ctypes.data_as(self._rp),
                                                 shp[0], shp[1], shp[2]))

    def has_field(self, name):
        """False for the radiation / TURB_* fields of a handle created without that switch."""
        return bool(self.lib.dycore_has_field(self._h, name.encode()))

    def field_shape(self, name):
        s = (C.c_int32 * 3)()
        self._check(self.lib.dycore_field_shape(self._h, name.encode(), C.byref(s)))
        return tuple(s)

    def upload(self, name, host):
        a = np.ascontiguousarray(np.asarray(host, dtype=self.dtype))
        if a.shape != self.field_shape(name):
            raise DycoreError('upload(%s): host shape %s != %s' % (name, a.shape, self.field_shape(name)))
        self._check(self.lib.dycore_upload(self._h, name.encode(), a.ctypes.data_as(self._rp)))

    def download(self, name, out=None):
        shp = self.field_shape(name)
        if out is None:
            out = np.empty(shp, dtype=self.dtype)
        if out.shape != shp or out.dtype != self.dtype or not out.flags.c_contiguous:
            raise DycoreError('download(%s): need a C-contiguous %s array of shape %s' % (name, self.dtype.name, shp))
        self._check(self.lib.dycore_download(self._h, name.encode(), out.ctypes.data_as(self._rp)))
        return out

    def field_ptr(self, name):
        dev, pitch, rows, lev = C.c_void_p(), C.c_int64(), C.c_int64(), C.c_int64()
        self._check(self.lib.dycore_field_ptr(self._h, name.encode(), C.byref(dev), C.byref(pitch),
                                              C.byref(rows), C.byref(lev)))
        return dev.value, pitch.value, rows.value, lev.value

    def copy_field(self, dst, src):
        self._check(self.lib.dycore_copy_field(self._h, dst.encode(), src.encode()))

    def exchange_bc(self, name):
        self._check(self.lib.dycore_exchange_bc(self._h, name.encode()))

    # ---- stages ----
    def continuity(self, dt):
        self._check(self.lib.dycore_continuity(self._h, float(dt)))

    def momentum(self):
        self._check(self.lib.dycore_momentum(self._h))

    def momentum_prepare(self):
        self._check(self.lib.dycore_momentum_prepare(self._h))

    def uflx_tendency(self):
        self._check(self.lib.dycore_uflx_tendency(self._h))

    def vflx_tendency(self):
        self._check(self.lib.dycore_vflx_tendency(self._h))

    def pott_tendency(self):
        self._check(self.lib.dycore_pott_tendency(self._h))

    def moist_tendency(self):
        self._check(self.lib.dycore_moist_tendency(self._h))

    def euler_forward(self, dt):
        self._check(self.lib.dycore_euler_forward(self._h, float(dt)))

    def primary_diag(self):
        self._check(self.lib.dycore_primary_diag(self._h))

    def secondary_diag_rho(self):
        self._check(self.lib.dycore_secondary_diag_rho(self._h))

    def secondary_diag(self):
        self._check(self.lib.dycore_secondary_diag(self._h))

    def compute_turbulence(self):
        self._check(self.lib.dycore_compute_turbulence(self._h))

    def radiation(self, solar_constant, lambdas_center, dlambdas):
        """dycore_radiation: longwave / shortwave columns -> fluxes, divergences, dPOTTdt_RAD."""
        lam = np.ascontiguousarray(lambdas_center, dtype=np.float64)
        dlam = np.ascontiguousarray(dlambdas, dtype=np.float64)
        if lam.shape != dlam.shape or lam.ndim != 1:
            raise ValueError('lambdas_center and dlambdas must be 1-D arrays of one length')
        dp = C.POINTER(C.c_double)
        self._check(self.lib.dycore_radiation(self._h, float(solar_constant), int(lam.size),
                                              lam.ctypes.data_as(dp), dlam.ctypes.data_as(dp)))

    def step_matsuno(self, dt, nsteps=1, with_rho=None):
        with_rho = self.i_turb if with_rho is None else with_rho
        self._check(self.lib.dycore_step_matsuno(self._h, float(dt), int(nsteps), int(bool(with_rho))))

    # ---- latitude bands: step in parts + halo rows <-> message buffers ----
    def step_begin(self, with_rho=None):
        with_rho = self.i_turb if with_rho is None else with_rho
        self._check(self.lib.dycore_step_begin(self._h, int(bool(with_rho))))

    # pieces of part 1 (include/dycore.h) between which the neighbour exchanges are hidden
    PART_MOM_INNER, PART_MOM_EDGE, PART_TEMP = 10, 11, 12
    PART_MOM_INNER_A, PART_MOM_INNER_B = 13, 14        # MOM_INNER in two halves around the edge piece (include/dycore.h)

    @property
    def overlap_parts(self):
        return bool(self.lib.dycore_overlap_parts(self._h))

    @property
    def flux_from_winds(self):
        """The fused step re-forms UFLX, VFLX from the winds (they are not stored): a band exchanges
        rows of UCOL, VCOL instead."""
        return bool(self.lib.dycore_flux_from_winds(self._h))

    @property
    def band_diag_fields(self):
        """Diagnostic fields whose halo rows a band receives from their owner with the POTT rows (never re-diagnosed)."""
        buf = C.create_string_buffer(256)
        self._check(self.lib.dycore_band_row_fields(self._h, buf, 256))
        return [n for n in buf.value.decode().split(',') if n]

    def substep_part(self, dt, substep, part):
        self._check(self.lib.dycore_substep_part(self._h, float(dt), int(substep), int(part)))

    def use_stream(self, stream_ptr):
        """Redirect subsequent launches to the HIP stream with this handle (None: back to the
        core's own stream)."""
        self._check(self.lib.dycore_use_stream(self._h, C.c_void_p(stream_ptr or None)))

    @property
    def stream_ptr(self):
        return int(self.lib.dycore_get_stream(self._h) or 0)

    def substep_field(self, substep, name):
        r = self.lib.dycore_substep_field(self._h, int(substep), name.encode())
        return r.decode() if r else name

    @staticmethod
    def _names(names):
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        return arr

    def rows_size(self, names):
        n = C.c_int64()
        self._check(self.lib.dycore_rows_size(self._h, len(names), self._names(names), C.byref(n)))
        return n.value

    def pack_rows(self, names, rows, dev_ptr):
        r = (C.c_int32 * len(rows))(*rows)
        self._check(self.lib.dycore_pack_rows(self._h, len(names), self._names(names), r, C.c_void_p(dev_ptr)))

    def unpack_rows(self, names, rows, dev_ptr):
        r = (C.c_int32 * len(rows))(*rows)
        self._check(self.lib.dycore_unpack_rows(self._h, len(names), self._names(names), r, C.c_void_p(dev_ptr)))

    # ---- measurement ----
    def time_steps(self, dt, nsteps, with_rho=None):
        with_rho = self.i_turb if with_rho is None else with_rho
        ms = C.c_double()
        self._check(self.lib.dycore_time_steps(self._h, float(dt), int(nsteps), int(bool(with_rho)), C.byref(ms)))
        return ms.value

    def profile(self, enable=True):
        """True / 1: HIP events around every launch; 2: around the dominant (momentum) kernel only."""
        self._check(self.lib.dycore_profile(self._h, int(enable)))

    def kernel_stats(self):
        out = {}
        for idx in range(self.lib.dycore_n_kernels()):
            name, n, ms = C.c_char_p(), C.c_int64(), C.c_double()
            self._check(self.lib.dycore_kernel_stats(self._h, idx, C.byref(name), C.byref(n), C.byref(ms)))
            out[name.value.decode()] = (n.value, ms.value)
        return out


UNIQUE_ID_BYTES = 128       # sizeof(ncclUniqueId)


def bands_unique_id(dtype=np.float64):
    """The 128-byte RCCL rendezvous id (rank 0 obtains it, every rank passes it to NativeBands.rccl).
    `dtype`: the build to ask (either one resolves the same librccl of the process)."""
    lib = load_library(dtype=dtype)
    buf = C.create_string_buffer(UNIQUE_ID_BYTES)
    rc = lib.dycore_bands_unique_id(buf, UNIQUE_ID_BYTES)
    if rc != 0:
        raise DycoreError('libdycore_hip: %s (code %d)' % (lib.dycore_last_error().decode(), rc))
    return buf.raw


class NativeBands:
    """The band step inside the library (include/dycore.h, dycore_bands_*): schedule, packing, streams, events and
    the transport -- RCCL send / recv, or device copies between the bands of one process -- behind one call per
    step sequence."""

    def __init__(self, lib, handle, cores):
        self.lib, self._b, self.cores = lib, handle, cores

    @classmethod
    def local(cls, cores):
        lib = cores[0].lib              # the build (working precision) the handles belong to
        arr = (C.c_void_p * len(cores))(*[c._h for c in cores])
        b = C.c_void_p()
        rc = lib.dycore_bands_create_local(arr, len(cores), C.byref(b))
        if rc != 0:
            raise DycoreError('libdycore_hip: %s (code %d)' % (lib.dycore_last_error().decode(), rc))
        return cls(lib, b, list(cores))

    @classmethod
    def rccl(cls, core, unique_id, rank, nranks, south, north):
        lib = core.lib
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError('the RCCL id has %d bytes' % UNIQUE_ID_BYTES)
        b = C.c_void_p()
        rc = lib.dycore_bands_create_rccl(core._h, C.c_char_p(bytes(unique_id)), UNIQUE_ID_BYTES, int(rank), int(nranks),
                                          -1 if south is None else int(south), -1 if north is None else int(north), C.byref(b))
        if rc != 0:
            raise DycoreError('libdycore_hip: %s (code %d)' % (lib.dycore_last_error().decode(), rc))
        return cls(lib, b, [core])

    def _check(self, rc):
        if rc != 0:
            raise DycoreError('libdycore_hip: %s (code %d)' % (self.lib.dycore_last_error().decode(), rc))

    def connect(self):
        self._check(self.lib.dycore_bands_connect(self._b))

    def step_matsuno(self, dt, nsteps=1, with_rho=None):
        with_rho = self.cores[0].i_turb if with_rho is None else with_rho
        self._check(self.lib.dycore_bands_step_matsuno(self._b, float(dt), int(nsteps), int(bool(with_rho))))

    def close(self):
        if getattr(self, '_b', None) and self._b.value:
            self.lib.dycore_bands_destroy(self._b)
            self._b = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
