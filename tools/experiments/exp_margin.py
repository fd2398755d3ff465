"""Parity margin of the fused step: max-norm relative error against the oracle after 1, 2, 10 steps
(4 deg x 8 levels golden initial state) and against the golden snapshots."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
from golden_util import load_golden, rel_max_err
from oracle_binding import oracle_from_golden
from hip_util import grid_from_golden, make_core, fields_from
for case in ('g4x8_dry', 'g10x4_dry'):
    z = load_golden(case)
    nx, ny, dt = int(z['meta/nx']), int(z['meta/ny']), float(z['meta/dt'])
    o = oracle_from_golden(z, None, i_turb=0); o.primary_diag()
    GR = grid_from_golden(z, 0); core = make_core(GR, 0)
    F = fields_from(GR, core, {k[3:]: z[k] for k in z.files if k.startswith('ic/')})
    core.primary_diag()
    done = 0
    for n in (1, 2, 10):
        core.step_matsuno(dt, n - done); o.step_matsuno(dt, n - done, with_rho=False); done = n
        line = []
        for name in ('COLP', 'UWIND', 'VWIND', 'POTT'):
            e, _, bad = rel_max_err(core.download(name), o.F[name], name, nx, ny)
            key = 'step%03d/%s' % (n, name)
            eg = rel_max_err(core.download(name), z[key], name, nx, ny)[0] if key in z.files else float('nan')
            line.append('%s %.1e/%.1e' % (name, e, eg))
        print(case, 'step', n, ' (vs oracle / vs golden) ', '  '.join(line), flush=True)
    core.close()
