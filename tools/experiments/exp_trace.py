"""Per-wavefront timeline of the whole-substep kernel (a -DDYC_TRACE=<block> build, tools: see profiles/r02_notes.md):
cycles between the trace points of every level, averaged over the levels of one workgroup.
usage: DYCORE_LIB=climate_model_amd/csrc/libexp_trace.so python tools/experiments/exp_trace.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.synthetic import initialize_fields
GR = Grid(Namelist(nz=64, dlat_deg=0.25, dlon_deg=0.25))
core = HipDycore(GR.nx, GR.ny, GR.nz)
core.set_grid(GR.GRF['HIP'])
F = ModelFields(GR, core); initialize_fields(GR, F); F.copy_host_to_device()
core.primary_diag(); core.step_matsuno(GR.dt, 3); core.sync()
buf = np.zeros((8, 256, 8), dtype=np.int64)
fn = core.lib.dycore_debug_trace
fn.argtypes = [C.c_void_p, C.c_void_p]
assert fn(core._h, buf.ctypes.data_as(C.c_void_p)) == 0
nz = 64
t = buf[:, :nz, :].astype(np.float64)          # [wave][level][point]; levels run nz-1 .. 0
names = ['momentum'] * 4 + ['temperature'] * 4
print('cycle counter ticks (s_memtime) per level, mean over levels 4..59 of one workgroup (the last launch: corrector)')
print('%-14s %9s %9s %9s %9s %9s %9s %9s' % ('wavefront', 'top:stores', 'top:DMA', 'arithm.', 'wait', 'end:mem/flux', 'barrier', 'level'))
for w in range(8):
    k = np.arange(59, 3, -1)
    mom = w < 4
    p4 = t[w, k, 4] if mom else t[w, k, 0]
    a0 = p4 - t[w, k, 0]
    a1 = t[w, k, 1] - p4
    b = t[w, k, 2] - t[w, k, 1]
    c0 = t[w, k, 5] - t[w, k, 2]
    c1 = t[w, k, 3] - t[w, k, 5]
    d = t[w, k - 1, 0] - t[w, k, 3]
    tot = t[w, k - 1, 0] - t[w, k, 0]
    print('%d %-12s %9.0f %9.0f %9.0f %9.0f %9.0f %9.0f %9.0f' % (w, names[w], a0.mean(), a1.mean(), b.mean(), c0.mean(), c1.mean(), d.mean(), tot.mean()))
k = np.arange(59, 3, -1)
print('arrival at the barrier relative to the first wavefront (mean ticks):', np.round((t[:, k, 3] - t[:, k, 3].min(axis=0)).mean(axis=1)).astype(int).tolist())
