"""MI355X-native hydrostatic dynamical-core Matsuno step (drop-in for the hot path of
Potopoles/Climate_Model).  See DESIGN.md and INTEGRATION.md."""
from .namelist import Namelist
from .grid import Grid
from .fields import ModelFields, FIELD_TABLE, CPU, HIP
from .hip_backend import HipDycore, DycoreError
from .discretizations import TendencyFactory, DiagnosticsFactory, PrognosticsFactory
from .matsuno import step_matsuno, step_matsuno_fused, compute_tendencies
from .turbulence import Turbulence
from .radiation import Radiation
from . import solver

__all__ = ['Namelist', 'Grid', 'ModelFields', 'FIELD_TABLE', 'CPU', 'HIP', 'HipDycore',
           'DycoreError', 'TendencyFactory', 'DiagnosticsFactory', 'PrognosticsFactory',
           'step_matsuno', 'step_matsuno_fused', 'compute_tendencies', 'Turbulence', 'Radiation']
