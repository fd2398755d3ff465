import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_terminal_summary(terminalreporter):
    """Largest max-norm relative error each test measured through golden_util.rel_max_err (the margin
    under its tolerance); also written to gpurun_out/parity_margins.txt when that directory exists."""
    try:
        from golden_util import ERR_LOG
    except Exception:
        return
    if not ERR_LOG:
        return
    worst = {}
    levels = []
    for node, name, err in ERR_LOG:
        if '(per level' in name:
            levels.append((node, name, err))
            continue
        if node not in worst or err > worst[node][1]:
            worst[node] = (name, err)
    lines = ['%-100s %-8s %.2e' % (node[-100:], name, err) for node, (name, err) in sorted(worst.items())]
    terminalreporter.write_sep('-', 'largest relative error per test (golden_util.rel_max_err)')
    for l in lines:
        terminalreporter.write_line(l)
    # the level-by-level norm (golden_util.rel_level_err): every field it was taken on
    lv = ['%-100s %-40s %.2e' % (node[-100:], name, err) for node, name, err in levels]
    if lv:
        terminalreporter.write_sep('-', 'per-level norm: max over k of max|d|_k / max|ref|_k (golden_util.rel_level_err)')
        for l in lv:
            terminalreporter.write_line(l)
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        try:
            with open(os.path.join(out, 'parity_margins.txt'), 'w') as fh:
                fh.write('\n'.join(lines) + '\n')
                if lv:
                    fh.write('\nper-level norm: max over k of max|d|_k / max|ref|_k, scale = the largest reference value of that level\n')
                    fh.write('\n'.join(lv) + '\n')
        except OSError:
            pass
