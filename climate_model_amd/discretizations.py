"""Operator boundary of the dycore: the drop-in for `dyn_org_discretizations.py`.

Same three factory classes, method names, argument lists and `fields_*` attributes as
the reference (TendencyFactory :75-294, DiagnosticsFactory :299-347,
PrognosticsFactory :351-393), so a driver written against the reference's calling
convention

    Tendencies.continuity(GR, GR.GRF[target], **F.get(Tendencies.fields_continuity, target=target))

runs unchanged.  The only target is 'HIP': the field arguments are `DeviceField`
handles naming buffers inside one `HipDycore` (libdycore_hip.so), every method is a
single C-ABI call, and the lateral exchanges the reference issues after each kernel
(`exchange_BC_*`) happen inside that call.  There is no CPU path here.
"""
from .fields import HIP, DeviceField
from .utilities import function_input_fields


def _core_of(*fields):
    core = None
    for f in fields:
        if f is None:
            continue
        if not isinstance(f, DeviceField):
            raise TypeError("target 'HIP' takes DeviceField handles (ModelFields.get(..., target='HIP')), got %r" % type(f))
        if core is None:
            core = f.core
        elif f.core is not core:
            raise ValueError('fields belong to different dycore handles')
    if core is None:
        raise ValueError('no device field passed')
    return core


def _expect(**named):
    """The C ABI addresses fields by name; a caller that passes a different field under a
    keyword (legal with numba arrays in the reference) is rejected instead of ignored."""
    for kw, f in named.items():
        if f is not None and f.name != kw:
            raise ValueError("field passed as %s is '%s'; the HIP target binds fields by name" % (kw, f.name))


class _Factory:
    def __init__(self, target=HIP):
        if target != HIP:
            raise ValueError("the only compute target of this package is 'HIP' (no CPU fallback)")
        self.target = target


class TendencyFactory(_Factory):

    def __init__(self, target=HIP):
        super().__init__(target)
        self.fields_continuity = function_input_fields(self.continuity)
        self.fields_momentum = function_input_fields(self.momentum)
        self.fields_temperature = function_input_fields(self.temperature)
        self.fields_moisture = function_input_fields(self.moisture)

    def continuity(self, GR, GRF, UFLX, VFLX, FLXDIV,
                   UWIND, VWIND, WWIND,
                   COLP, dCOLPdt, COLP_NEW, COLP_OLD):
        _expect(UFLX=UFLX, VFLX=VFLX, FLXDIV=FLXDIV, UWIND=UWIND, VWIND=VWIND, WWIND=WWIND,
                COLP=COLP, dCOLPdt=dCOLPdt, COLP_NEW=COLP_NEW, COLP_OLD=COLP_OLD)
        _core_of(UFLX, UWIND, COLP).continuity(GR.dt)

    def momentum(self, GRF,
                 dUFLXdt, dVFLXdt,
                 UWIND, VWIND, WWIND,
                 UFLX, VFLX,
                 CFLX, QFLX, DFLX, EFLX,
                 SFLX, TFLX, BFLX, RFLX,
                 PHI, PHIVB, COLP, COLP_NEW, POTT,
                 PVTF, PVTFVB,
                 WWIND_UWIND, WWIND_VWIND,
                 KMOM_dUWINDdz, KMOM_dVWINDdz,
                 KMOM, RHOVB, RHO,
                 dUFLXdt_TURB, dVFLXdt_TURB,
                 SMOMXFLX, SMOMYFLX):
        _expect(dUFLXdt=dUFLXdt, dVFLXdt=dVFLXdt, UWIND=UWIND, VWIND=VWIND, WWIND=WWIND,
                UFLX=UFLX, VFLX=VFLX, COLP=COLP, COLP_NEW=COLP_NEW, POTT=POTT)
        _core_of(dUFLXdt, dVFLXdt, UWIND).momentum()

    def temperature(self, GRF,
                    dPOTTdt, POTT, UFLX, VFLX,
                    COLP, POTTVB, WWIND, COLP_NEW,
                    PHI, PHIVB, KHEAT, RHO, RHOVB, SSHFLX,
                    dPOTTdt_TURB, dPOTTdt_RAD):
        _expect(dPOTTdt=dPOTTdt, POTT=POTT, UFLX=UFLX, VFLX=VFLX, COLP=COLP, POTTVB=POTTVB,
                WWIND=WWIND, COLP_NEW=COLP_NEW)
        _core_of(dPOTTdt, POTT).pott_tendency()

    def moisture(self, GRF,
                 dQVdt, dQVdt_TURB, QV, dQCdt, QC, UFLX, VFLX,
                 COLP, WWIND, COLP_NEW,
                 PHI, PHIVB, KHEAT, RHO, RHOVB, SLHFLX):
        """dQVdt, dQCdt (dyn_moist.py) on handles created with i_moist; a no-op otherwise, so that
        the reference's compute_tendencies sequence (dyn_tendencies.py:62-70) runs unchanged on
        the dry dycore (QV/QC do not feed COLP/UWIND/VWIND/POTT)."""
        core = _core_of(QV, QC, UFLX)
        if getattr(core, 'i_moist', False):
            _expect(dQVdt=dQVdt, dQVdt_TURB=dQVdt_TURB, QV=QV, dQCdt=dQCdt, QC=QC)
            core.moist_tendency()


class DiagnosticsFactory(_Factory):

    def __init__(self, target=HIP):
        super().__init__(target)
        self.fields_primary_diag = function_input_fields(self.primary_diag)
        self.fields_secondary_diag = function_input_fields(self.secondary_diag)

    def primary_diag(self, GRF,
                     COLP, PVTF, PVTFVB,
                     PHI, PHIVB, POTT, POTTVB, HSURF):
        _expect(COLP=COLP, PVTF=PVTF, PVTFVB=PVTFVB, PHI=PHI, PHIVB=PHIVB, POTT=POTT,
                POTTVB=POTTVB, HSURF=HSURF)
        _core_of(COLP, PVTF, POTT).primary_diag()

    def secondary_diag(self,
                       POTTVB, TAIRVB, PVTFVB,
                       COLP, PAIR, PAIRVB, PHI, POTT,
                       TAIR, RHO, RHOVB, PVTF,
                       UWIND, VWIND, WINDX, WINDY, WIND):
        """dyn_org_discretizations.py:329-347, every output of diag_secondary
        (dyn_diagnostics.py:199-222)."""
        _expect(POTTVB=POTTVB, TAIRVB=TAIRVB, PVTFVB=PVTFVB, PAIR=PAIR, PAIRVB=PAIRVB, POTT=POTT,
                TAIR=TAIR, RHO=RHO, RHOVB=RHOVB, PVTF=PVTF, UWIND=UWIND, VWIND=VWIND,
                WINDX=WINDX, WINDY=WINDY, WIND=WIND)
        _core_of(RHO, RHOVB, WIND).secondary_diag()


class PrognosticsFactory(_Factory):

    def __init__(self, target=HIP):
        super().__init__(target)
        self.fields_prognostic = function_input_fields(self.euler_forward)

    def euler_forward(self, GR, GRF, UWIND_OLD, UWIND, VWIND_OLD,
                      VWIND, COLP_OLD, COLP, POTT_OLD, POTT,
                      QV, QV_OLD, QC, QC_OLD,
                      dUFLXdt, dVFLXdt, dPOTTdt, dQVdt, dQCdt):
        _expect(UWIND_OLD=UWIND_OLD, UWIND=UWIND, VWIND_OLD=VWIND_OLD, VWIND=VWIND,
                COLP_OLD=COLP_OLD, COLP=COLP, POTT_OLD=POTT_OLD, POTT=POTT,
                dUFLXdt=dUFLXdt, dVFLXdt=dVFLXdt, dPOTTdt=dPOTTdt)
        _core_of(UWIND, VWIND, POTT).euler_forward(GR.dt)
