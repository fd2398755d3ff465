"""Run configuration of the dycore hot path.

The reference keeps its configuration as module globals of `namelist.py` that
numba freezes into the kernels (namelist.py:30-316, io_read_namelist.py:30-92).
Here the same names live on a `Namelist` object that is handed to `Grid`; the
values that reach the device travel in `DycoreConfig` (include/dycore.h).
Only the entries the Matsuno step uses are kept.
"""
from dataclasses import dataclass, field
from datetime import datetime


# namelist.py:286-316: UVFLX diffusion coefficient by resolution, then doubled
_UVFLX_DIF_TABLE = {10: 1.5, 8: 1.5, 6: 1.8, 5: 2, 4: 2.5, 3: 3.3, 2: 5, 1.5: 7.5}


def uvflx_dif_coef_for(dlat_deg):
    if dlat_deg in _UVFLX_DIF_TABLE:
        c = _UVFLX_DIF_TABLE[dlat_deg]
    elif dlat_deg <= 1:
        c = 10
    else:
        c = 0
    return c * 2.0


# include/dycore.h DYCORE_TERM_*: bit of DycoreConfig.term_off <- the namelist entry that, set to 0, switches the term off
TERM_BITS = {
    'i_COLP_main_switch': 1 << 0,
    'i_UVFLX_main_switch': 1 << 1, 'i_UVFLX_hor_adv': 1 << 2, 'i_UVFLX_vert_adv': 1 << 3, 'i_UVFLX_coriolis': 1 << 4,
    'i_UVFLX_pre_grad': 1 << 5, 'i_UVFLX_num_dif': 1 << 6,
    'i_POTT_main_switch': 1 << 7, 'i_POTT_hor_adv': 1 << 8, 'i_POTT_vert_adv': 1 << 9, 'i_POTT_num_dif': 1 << 10,
    'i_moist_hor_adv': 1 << 11, 'i_moist_vert_adv': 1 << 12, 'i_moist_num_dif': 1 << 13,
}


@dataclass
class Namelist:
    # grid (namelist.py:30-40, 243-278)
    nz: int = 32
    nb: int = 1
    lon0_deg: float = 0.
    lon1_deg: float = 360.
    lat0_deg: float = -90.
    lat1_deg: float = 90.
    dlat_deg: float = 1.0
    dlon_deg: float = 1.0
    pair_top: float = 10000.
    # dynamics switches (namelist.py:62-88).  The dry dycore runs with the
    # turbulence/radiation couplings off (their inputs are identically zero).
    i_UVFLX_vert_turb: int = 0
    i_POTT_vert_turb: int = 0
    # per-term switches (namelist.py:62-88, all 1 in the reference's shipped file).  A run with any of them at 0 steps on
    # the stage kernels, where the reference consumes them (dyn_UFLX.py:117-193, dyn_VFLX.py:115-192, dyn_POTT.py:72-108,
    # dyn_moist.py:68-115, dyn_continuity.py:211, dyn_UVFLX_prepare.py:260-356); they travel as DycoreConfig.term_off
    i_COLP_main_switch: int = 1
    i_UVFLX_main_switch: int = 1
    i_UVFLX_hor_adv: int = 1
    i_UVFLX_vert_adv: int = 1
    i_UVFLX_coriolis: int = 1
    i_UVFLX_num_dif: int = 1
    i_UVFLX_pre_grad: int = 1
    i_POTT_main_switch: int = 1
    i_POTT_hor_adv: int = 1
    i_POTT_vert_adv: int = 1
    i_POTT_num_dif: int = 1
    i_moist_hor_adv: int = 1
    i_moist_vert_adv: int = 1
    i_moist_num_dif: int = 1
    i_POTT_radiation: int = 0
    i_moist_main_switch: int = 0        # dyn_moist: QV/QC advection (namelist.py:80)
    # QV, QC < 0 -> 0 after every Euler step: what the reference's accelerator kernel does (dyn_timestep.py:145-154,
    # its default i_comp_mode = 2); 0 = its CPU kernel (:291-296), the path the parity fixtures come from
    i_moist_clip: int = 0
    i_radiation: int = 0                # radiation scheme (namelist.py:104); implies i_POTT_radiation
    rad_nth_hour: float = 3.9           # namelist.py:117
    GMT_initialization: datetime = field(default_factory=lambda: datetime(2019, 1, 1, 0, 0, 0))   # namelist.py:21
    # diffusion (namelist.py:281-316); None = table value for dlat_deg
    UVFLX_dif_coef: float = None
    POTT_dif_coef: float = 1E-5
    moist_dif_coef: float = None        # None = POTT_dif_coef (namelist.py:284)
    # time stepping (namelist.py:219-221)
    CFL: float = 0.7
    dt: float = None            # None = reference rule int(CFL*min(dx)/400), lowered until it divides the output interval
    # run length and output / restart intervals (namelist.py:147, 243-278; values of the shipped preset 2).
    # i_out_nth_hour also enters the time-step rule (main_grid.py:262-263).
    i_sim_n_days: float = 365.0
    i_out_nth_hour: float = 12.0
    i_restart_nth_day: float = 1.0
    nth_ts_print_diag: int = 50         # namelist.py:145
    working_precision: str = 'float64'

    def __post_init__(self):
        if self.nb != 1:
            raise NotImplementedError('nb > 1 not implemented (io_read_namelist.py:30-31)')
        if self.lon0_deg != 0 or self.lon1_deg != 360:
            raise NotImplementedError('only periodic longitude 0..360 (io_read_namelist.py:32-34)')
        if self.working_precision not in ('float64', 'float32'):
            raise ValueError("working_precision is 'float64' or 'float32' (io_read_namelist.py:47-53)")
        if bool(self.i_UVFLX_vert_turb) != bool(self.i_POTT_vert_turb):
            raise NotImplementedError('i_UVFLX_vert_turb and i_POTT_vert_turb must agree')
        if self.i_radiation:
            self.i_POTT_radiation = 1
        if self.moist_dif_coef is None:
            self.moist_dif_coef = self.POTT_dif_coef
        if self.UVFLX_dif_coef is None:
            self.UVFLX_dif_coef = uvflx_dif_coef_for(self.dlat_deg)

    @property
    def wp(self):
        """numpy dtype of the fields (io_read_namelist.py:47-53).  The grid constants are formed in float64 and cast
        when they are handed to a float32 handle."""
        import numpy as np
        return np.dtype(self.working_precision)

    @property
    def term_off(self):
        """DycoreConfig.term_off: one bit per namelist entry of TERM_BITS that is 0."""
        m = 0
        for name, bit in TERM_BITS.items():
            if not getattr(self, name):
                m |= bit
        return m

    @property
    def i_turb(self):
        return int(bool(self.i_UVFLX_vert_turb))
