"""Matsuno (Euler-backward) time integration of the dycore.

`step_matsuno(GR, F)` reproduces the reference's driver line by line in structure
(dyn_matsuno.py:28-131 and dyn_tendencies.py:25-70): save *_OLD, then twice
{tendencies, COLP <- COLP_NEW, Euler forward, primary diagnostics}, each through the
factory boundary, one C-ABI call per stage.  `step_matsuno_fused(GR, F, nsteps)` is
the production path: the same sequence issued by libdycore_hip.so itself
(dycore_step_matsuno) with the *_OLD copies replaced by buffer rotation.
"""
from .fields import HIP
from .discretizations import TendencyFactory, DiagnosticsFactory, PrognosticsFactory

Tendencies = TendencyFactory(target=HIP)
Prognostics = PrognosticsFactory(target=HIP)
Diagnostics = DiagnosticsFactory(target=HIP)

_OLD_PAIRS = (('COLP_OLD', 'COLP'), ('UWIND_OLD', 'UWIND'), ('VWIND_OLD', 'VWIND'), ('POTT_OLD', 'POTT'))


def compute_tendencies(GR, F):
    """dyn_tendencies.py:25-70"""
    GR.timer.start('cont')
    Tendencies.continuity(GR, GR.GRF[Tendencies.target],
                          **F.get(Tendencies.fields_continuity, target=Tendencies.target))
    GR.timer.stop('cont')

    GR.timer.start('wind')
    Tendencies.momentum(GR.GRF[Tendencies.target],
                        **F.get(Tendencies.fields_momentum, target=Tendencies.target))
    GR.timer.stop('wind')

    GR.timer.start('temp')
    Tendencies.temperature(GR.GRF[Tendencies.target],
                           **F.get(Tendencies.fields_temperature, target=Tendencies.target))
    GR.timer.stop('temp')

    GR.timer.start('moist')
    Tendencies.moisture(GR.GRF[Tendencies.target],
                        **F.get(Tendencies.fields_moisture, target=Tendencies.target))
    GR.timer.stop('moist')


def step_matsuno(GR, F):
    """dyn_matsuno.py:28-131, stage by stage through the factory boundary."""
    core = F.core
    GR.timer.start('step')
    for old, cur in _OLD_PAIRS:                      # :34-49 (set_equal)
        core.copy_field(old, cur)
    if getattr(core, 'i_moist', False):
        core.copy_field('QV_OLD', 'QV')
        core.copy_field('QC_OLD', 'QC')
    GR.timer.stop('step')

    for _substep in ('estimate', 'final'):
        compute_tendencies(GR, F)                    # :63 / :102
        core.copy_field('COLP', 'COLP_NEW')          # :64-67 / :103-106

        GR.timer.start('step')
        Prognostics.euler_forward(GR, GR.GRF[Prognostics.target],
                                  **F.get(Prognostics.fields_prognostic, target=Prognostics.target))
        GR.timer.stop('step')

        GR.timer.start('diag')
        Diagnostics.primary_diag(GR.GRF[Diagnostics.target],
                                 **F.get(Diagnostics.fields_primary_diag, target=Diagnostics.target))
        GR.timer.stop('diag')
    GR.ts += 1


def step_matsuno_fused(GR, F, nsteps=1, with_rho=None):
    """The whole hot loop body inside the library: dycore_step_matsuno.  with_rho=False: RHO / RHOVB are
    current (the caller has just run secondary_diag, as solver.py:99-101 does before every step)."""
    GR.timer.start('step')
    F.core.step_matsuno(GR.dt, nsteps, with_rho)
    GR.timer.stop('step')
    GR.ts += nsteps
