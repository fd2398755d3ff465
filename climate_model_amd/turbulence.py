"""Turbulence scheme of the reference behind its own interface.

Counterpart of `turb_main.Turbulence` (turb_main.py:30-48): bulk-Richardson K coefficients
KMOM, KHEAT (turb_compute.py:54-142) at the layer interfaces, column-local, consumed by the
turbulent terms of the dycore (`i_UVFLX_vert_turb`, `i_POTT_vert_turb`, moisture).
"""
from .fields import HIP
from .discretizations import _core_of, _expect
from .utilities import function_input_fields


class Turbulence:

    def __init__(self, GR=None, target=HIP):
        if target != HIP:
            raise ValueError("the only compute target of this package is 'HIP'")
        self.target = target
        self.fields_main = function_input_fields(self.compute_turbulence)

    def compute_turbulence(self, GR, KMOM, KHEAT, PHIVB, HSURF, PHI, QV,
                           WINDX, WINDY, POTTVB, POTT):
        _expect(KMOM=KMOM, KHEAT=KHEAT, PHIVB=PHIVB, HSURF=HSURF, PHI=PHI, QV=QV,
                WINDX=WINDX, WINDY=WINDY, POTTVB=POTTVB, POTT=POTT)
        _core_of(KMOM, KHEAT).compute_turbulence()
