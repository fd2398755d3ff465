"""Time loop around the dycore step, the way the reference's `solver.py:82-226` runs it.

Per time step, in the reference's order: status line every `nth_ts_print_diag` steps
(io_functions.py:93-134, with the "MODEL CRASH" test), secondary diagnostics, turbulence and
radiation if switched on, then the Matsuno step.  Microphysics, the surface scheme and NetCDF output of the reference's loop are
outside the hot path (out of scope) and have no slot here; the fields they would
produce (QC, SOILTEMP, surface albedos) are inputs.

`fused_steps > 1` hands up to that many Matsuno steps to the library in one call
(dycore_step_matsuno) when no per-step physics is active; chunks end at the status lines.
"""
from datetime import timedelta

import numpy as np

from .fields import HIP
from .discretizations import DiagnosticsFactory
from .matsuno import step_matsuno, step_matsuno_fused
from .turbulence import Turbulence
from .radiation import Radiation

Diagnostics = DiagnosticsFactory(target=HIP)


def diagnose_print_diag_fields(GR, WIND, COLP, POTT):
    """io_functions.py:68-90: max wind, mass-weighted mean wind and potential temperature,
    area-weighted mean column pressure."""
    ii, jj = GR.ii, GR.jj
    A = GR.A
    w = COLP[ii, jj, 0] * A[ii, jj, 0]
    max_wind = np.max(WIND[ii, jj, :])
    mean_wind = 0
    mean_temp = 0
    for k in range(GR.nz):
        mean_wind += np.sum(WIND[ii, jj, k] * w) / np.sum(w)
        mean_temp += np.sum(POTT[ii, jj, k] * w) / np.sum(w)
    mean_wind = mean_wind / GR.nz
    mean_temp = mean_temp / GR.nz
    mean_colp = np.sum(COLP[ii, jj, 0] * A[ii, jj, 0]) / np.sum(A[ii, jj, 0])
    return max_wind, mean_wind, mean_temp, mean_colp


def print_ts_info(GR, F, nth_ts_print_diag=50, out=print):
    """io_functions.py:93-111.  Returns the four diagnostics (or None when it is not a
    printing step)."""
    if GR.ts % nth_ts_print_diag != 0:
        return None
    F.copy_device_to_host(GR, F.PRINT_DIAG_FIELDS)
    GR.timer.start('diag')
    vmax, mean_wind, mean_temp, mean_colp = diagnose_print_diag_fields(
        GR, F.host['WIND'], F.host['COLP'], F.host['POTT'])
    GR.timer.stop('diag')
    out(str(GR.ts) + '  ' + str(np.round(GR.ts * GR.dt / 3600 / 24, 3)) + '\t days' +
        ' vmax: ' + str(np.round(vmax, 1)) + '  m/s vmean: ' + str(np.round(mean_wind, 3)) +
        ' m/s Tmean: ' + str(np.round(mean_temp, 7)) + '  K  COLP: ' + str(np.round(mean_colp, 2)) + ' Pa')
    UW = F.host['UWIND'][GR.iis, GR.jj]
    if np.sum(np.isnan(UW)) > 0 or np.max(UW) > 500:
        raise ValueError('MODEL CRASH')
    return vmax, mean_wind, mean_temp, mean_colp


def run(GR, F, nts, nth_ts_print_diag=None, fused=True, fused_steps=1, out=print,
        output_path=None, i_out_nth_hour=None, restart_dir=None, i_restart_nth_day=None):
    """solver.py:70-226.  F.host must already be uploaded.  Returns the list of printed
    diagnostics tuples.  With `output_path`, history files out%04d.nc are written every
    `i_out_nth_hour` simulated hours (and constants.nc at the start) as io_nc_output.py does; with
    `restart_dir`, a restart archive every `i_restart_nth_day` days (solver.py:156-177).  The
    intervals default to the namelist's (GR.nl), the step counts to main_grid.py:264-270.

    Order inside one iteration as in the reference: status line, secondary diagnostics, turbulence,
    radiation, Matsuno step, output.  The status line therefore shows the WIND the secondary
    diagnostics of the PREVIOUS iteration left (the state before that iteration's dynamics), next
    to the current COLP / POTT -- solver.py:91 comes before :98-101.  Before the first iteration WIND
    holds the diagnosis of the initial state (io_initial_conditions.py:101-104)."""
    core = F.core
    nl = GR.nl
    if nth_ts_print_diag is None:
        nth_ts_print_diag = getattr(nl, 'nth_ts_print_diag', 50)
    turb = Turbulence(GR) if nl.i_turb else None
    rad = Radiation(GR, nl.rad_nth_hour) if nl.i_radiation else None
    fused_steps = max(1, int(fused_steps))
    if turb is not None or rad is not None or not fused:
        fused_steps = 1
    log = []

    GR.timer.start('diag')
    Diagnostics.primary_diag(GR.GRF[Diagnostics.target],
                             **F.get(Diagnostics.fields_primary_diag, target=Diagnostics.target))
    if GR.ts == 0:      # WIND (PAIR, TAIR, RHO) of the initial state, as initialize_fields leaves them
        Diagnostics.secondary_diag(**F.get(Diagnostics.fields_secondary_diag, target=Diagnostics.target))
    GR.timer.stop('diag')
    if rad is not None and GR.ts == 0:
        rad.calc_radiation(GR, F)       # main_fields.py:94-101: a first radiation call on the initial state

    out_nth_ts = rst_nth_ts = 0
    if output_path is not None:
        from . import io_output
        hours = i_out_nth_hour if i_out_nth_hour is not None else getattr(nl, 'i_out_nth_hour', 12.)
        out_nth_ts = max(1, int(hours * 3600 / GR.dt))                           # main_grid.py:268
        io_output.constant_fields_to_NC(GR, F, output_path)                      # solver.py:66
        if not hasattr(GR, 'nc_output_count'):
            GR.nc_output_count = 0
    if restart_dir is not None:
        from . import io_output
        hours = i_out_nth_hour if i_out_nth_hour is not None else getattr(nl, 'i_out_nth_hour', 12.)
        days = i_restart_nth_day if i_restart_nth_day is not None else getattr(nl, 'i_restart_nth_day', 1.)
        rst_nth_ts = max(1, int(days * 24 / hours * max(1, int(hours * 3600 / GR.dt))))      # main_grid.py:269-270
    if out_nth_ts or rst_nth_ts:
        fused_steps = 1

    end = GR.ts + int(nts)
    while GR.ts < end:
        GR.timer.start('total')
        step = GR.ts + 1                      # the reference counts the step before running it
        # a fused chunk runs one secondary-diagnostics call for all its steps, so it may contain neither a
        # status-line step nor the step before one (whose diagnostics that line shows) except as its first
        n = 1
        while (n < fused_steps and step + n <= end and (step + n) % nth_ts_print_diag != 0
               and (step + n + 1) % nth_ts_print_diag != 0):
            n += 1
        GR.GMT += timedelta(seconds=GR.dt * n)           # solver.py:88-90: clock first, then physics and dynamics

        if step % nth_ts_print_diag == 0:     # solver.py:91
            GR.ts = step
            try:
                log.append(print_ts_info(GR, F, nth_ts_print_diag, out))
            finally:
                GR.ts = step - 1

        GR.timer.start('diag')
        Diagnostics.secondary_diag(**F.get(Diagnostics.fields_secondary_diag, target=Diagnostics.target))
        GR.timer.stop('diag')

        if turb is not None:
            GR.timer.start('turb')
            turb.compute_turbulence(GR, **F.get(turb.fields_main, target=turb.target))
            GR.timer.stop('turb')

        if rad is not None:                   # solver.py:146-160 (surface scheme and microphysics, the two
            GR.ts = step                      # slots before it, are inputs here: SOILTEMP, albedos, QC)
            try:
                rad.launch_radiation_calc(GR, F)
            finally:
                GR.ts = step - 1

        if fused:
            step_matsuno_fused(GR, F, n, with_rho=False if n == 1 else None)     # secondary_diag above left RHO, RHOVB current
        else:
            step_matsuno(GR, F)

        if out_nth_ts and GR.ts % out_nth_ts == 0:      # solver.py:152-164: the counter first, then the file
            GR.timer.start('IO')
            io_output.output_to_NC(GR, F, output_path, count=GR.nc_output_count + 1, refresh_diag=False)
            GR.timer.stop('IO')
        if rst_nth_ts and GR.ts % rst_nth_ts == 0:
            GR.timer.start('IO')
            io_output.write_restart(GR, F, restart_dir)
            GR.timer.stop('IO')
        GR.timer.stop('total')
    core.sync()
    return log
