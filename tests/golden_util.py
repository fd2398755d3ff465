"""Helpers shared by the golden-vector tests (oracle and HIP path alike)."""
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

STAGE_ORDER = ['continuity', 'momentum', 'temperature', 'moisture', 'euler_forward', 'primary_diag']


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + '.npz'))


def stage_tags(z):
    return [str(t) for t in z['meta/stage_seq']]


def fields_of(z, tag):
    pre = tag + '/'
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


def compare_region(name, nx, ny):
    """Index region (i-slice, j-slice) in which a stage output is meaningful.

    Cells outside are ones the reference computes from never-written halos and
    discards at the next exchange_BC: dUFLXdt column nx+1,
    dVFLXdt rows 1 and ny+1.
    """
    if name in ('dUFLXdt', 'dUFLXdt_TURB'):
        return slice(1, nx + 1), slice(1, ny + 1)
    if name in ('dVFLXdt', 'dVFLXdt_TURB'):
        return slice(1, nx + 1), slice(2, ny + 1)
    if name == 'dCOLPdt':       # owned cells: nobody reads its halo (NaN in the reference; 0 with i_COLP_main_switch = 0,
        return slice(1, nx + 1), slice(1, ny + 1)      # where dyn_continuity.py:214 assigns the whole array)
    return slice(None), slice(None)


# every comparison a test makes: (pytest node id, field, error) -- conftest.py prints the largest error per test
# at the end of the session next to nothing else, so the margin under each stated tolerance is on record
ERR_LOG = []


def rel_max_err(test, ref, name, nx, ny):
    """max|test-ref| / max|ref| over cells where the reference is finite
    (the reference's own testsuite.py:46-60 norm).  Returns (err, n_compared,
    n_nonfinite_in_test)."""
    si, sj = compare_region(name, nx, ny)
    t = np.asarray(test)[si, sj]
    r = np.asarray(ref)[si, sj]
    m = np.isfinite(r)
    if not m.any():
        return 0.0, 0, 0
    bad = int((~np.isfinite(t[m])).sum())
    scale = np.max(np.abs(r[m]))
    d = np.max(np.abs(np.where(np.isfinite(t[m]), t[m], 0.0) - r[m]))
    err = d / scale if scale > 0 else d
    ERR_LOG.append((os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0], name, float(err)))
    return err, int(m.sum()), bad


def rel_level_err(test, ref, name, nx, ny):
    """The same norm taken LEVEL BY LEVEL: max over k of max|test-ref|_k / max|ref|_k (scale = the largest reference
    value of THAT level), so that small-magnitude regions of a field (WWIND aloft, QC where it lives at 1e-10, the
    winds of quiet levels) are not hidden behind the field's global maximum.  A level whose reference is identically
    zero (WWIND at the model top and at the surface) must be reproduced exactly.  Returns (err, worst level); 2-D
    fields fall back to rel_max_err."""
    si, sj = compare_region(name, nx, ny)
    t = np.asarray(test)[si, sj]
    r = np.asarray(ref)[si, sj]
    if r.ndim < 3 or r.shape[2] == 1:
        return rel_max_err(test, ref, name, nx, ny)[0], 0
    worst, kworst = 0.0, 0
    for k in range(r.shape[2]):
        rk, tk = r[:, :, k], t[:, :, k]
        m = np.isfinite(rk)
        if not m.any():
            continue
        if not np.isfinite(tk[m]).all():
            worst, kworst = np.inf, k
            break
        scale = np.max(np.abs(rk[m]))
        d = np.max(np.abs(tk[m] - rk[m]))
        e = d / scale if scale > 0 else (0.0 if d == 0 else np.inf)
        if e > worst:
            worst, kworst = float(e), k
    ERR_LOG.append((os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0], name + ' (per level, worst k=%d)' % kworst, float(worst)))
    return worst, kworst


# bit of DycoreConfig.term_off / oracle_cfg.term_off per namelist entry (include/dycore.h DYCORE_TERM_*); restated here so
# that the CPU tests of the oracle do not depend on the product package
TERM_BITS = {
    'i_COLP_main_switch': 1 << 0,
    'i_UVFLX_main_switch': 1 << 1, 'i_UVFLX_hor_adv': 1 << 2, 'i_UVFLX_vert_adv': 1 << 3, 'i_UVFLX_coriolis': 1 << 4,
    'i_UVFLX_pre_grad': 1 << 5, 'i_UVFLX_num_dif': 1 << 6,
    'i_POTT_main_switch': 1 << 7, 'i_POTT_hor_adv': 1 << 8, 'i_POTT_vert_adv': 1 << 9, 'i_POTT_num_dif': 1 << 10,
    'i_moist_hor_adv': 1 << 11, 'i_moist_vert_adv': 1 << 12, 'i_moist_num_dif': 1 << 13,
}


def term_off_of(z):
    """term_off mask of a golden case: the namelist entries its reference run had at 0 (meta/terms_off)."""
    m = 0
    if 'meta/terms_off' in z.files:
        for name in z['meta/terms_off']:
            m |= TERM_BITS[str(name)]
    return m
