"""Probe (GPU box): a dycore handle first, torch's device context afterwards, in one process.
Prints which libamdhip64 / libhsa-runtime64 files are mapped and whether torch.cuda came up.
    python tools/probe_runtime_order.py            # hip_backend._one_hip_runtime imports torch before mapping the library
    DYCORE_NO_TORCH_PRELOAD=1 python tools/probe_runtime_order.py    # the library (system runtime) first, torch afterwards
Run it under `timeout`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from climate_model_amd import HipDycore

core = HipDycore(36, 18, 4)
core.close()


def mapped():
    out = set()
    for line in open('/proc/self/maps'):
        f = line.split()[-1]
        if 'libamdhip64' in f or 'libhsa-runtime64' in f:
            out.add(f)
    return sorted(out)


print('after the handle:', mapped())
import torch
try:
    x = torch.zeros(4, device='cuda')
    print('torch.cuda ok:', float(x.sum()))
except Exception as e:
    print('torch.cuda FAILED:', repr(e)[:200])
print('after torch:', mapped())
