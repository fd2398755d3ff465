"""CPU-only checks of the C-ABI library and the host-side mirror of the reference's
operator boundary (no compute calls: there is no GPU here)."""
import inspect
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from climate_model_amd import hip_backend as hb
from climate_model_amd import (Namelist, Grid, ModelFields, TendencyFactory, DiagnosticsFactory,
                               PrognosticsFactory, FIELD_TABLE)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'climate_model_amd', 'csrc', 'libdycore_hip.so')


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'dycore.h')).read()
    declared = sorted(set(re.findall(r'\b(dycore_[a-z_]+)\s*\(', hdr)))
    assert declared, 'no declarations parsed'
    lib = hb.load_library()
    missing = [s for s in declared if not hasattr(lib, s)]
    assert missing == []
    assert sorted(hb.ABI_SYMBOLS) == declared
    assert lib.dycore_abi_version() == hb.ABI_VERSION


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    with pytest.raises(hb.DycoreError):
        hb.HipDycore(36, 18, 4)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, 'climate_model_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in txt.lower(), \
                    '%s mentions the oracle' % os.path.join(dirpath, f)


def test_factories_mirror_reference_signatures():
    """Method names and argument lists of dyn_org_discretizations.py:89-393."""
    T, D, P = TendencyFactory(), DiagnosticsFactory(), PrognosticsFactory()
    assert T.fields_continuity == ['UFLX', 'VFLX', 'FLXDIV', 'UWIND', 'VWIND', 'WWIND',
                                   'COLP', 'dCOLPdt', 'COLP_NEW', 'COLP_OLD']
    assert T.fields_momentum[:7] == ['dUFLXdt', 'dVFLXdt', 'UWIND', 'VWIND', 'WWIND', 'UFLX', 'VFLX']
    assert T.fields_momentum[-2:] == ['SMOMXFLX', 'SMOMYFLX'] and len(T.fields_momentum) == 33
    assert T.fields_temperature == ['dPOTTdt', 'POTT', 'UFLX', 'VFLX', 'COLP', 'POTTVB', 'WWIND',
                                    'COLP_NEW', 'PHI', 'PHIVB', 'KHEAT', 'RHO', 'RHOVB', 'SSHFLX',
                                    'dPOTTdt_TURB', 'dPOTTdt_RAD']
    assert P.fields_prognostic[:8] == ['UWIND_OLD', 'UWIND', 'VWIND_OLD', 'VWIND', 'COLP_OLD', 'COLP',
                                       'POTT_OLD', 'POTT']
    assert D.fields_primary_diag == ['COLP', 'PVTF', 'PVTFVB', 'PHI', 'PHIVB', 'POTT', 'POTTVB', 'HSURF']
    with pytest.raises(ValueError):
        TendencyFactory(target='CPU')


def test_grid_and_fields_shapes():
    GR = Grid(Namelist(nz=8, dlat_deg=4, dlon_deg=4))
    assert (GR.nx, GR.ny, GR.nz, GR.dt) == (90, 45, 8, 27)      # BASELINE config 0
    F = ModelFields(GR)
    assert F.host['UWIND'].shape == (93, 47, 8) and F.host['VWIND'].shape == (92, 48, 8)
    assert F.host['CFLX'].shape == (93, 48, 8) and F.host['WWIND'].shape == (92, 47, 9)
    assert np.isnan(F.host['UFLX']).all() and (F.host['WWIND'] == 0).all()
    for name in GR.GRID_FIELD_NAMES:
        assert name in GR.GRF['HIP']


@pytest.mark.parametrize('case,d,nz', [('g10x4_dry', 10, 4), ('g4x8_dry', 4, 8)])
def test_reference_initial_conditions_rebuilt(case, d, nz):
    """initial_conditions.py rebuilds the reference's sigma levels and initial state from its two
    input data files: compared with what the reference itself produced (golden ic/ and grid/)."""
    from climate_model_amd import initial_conditions as ic
    data_dir = os.path.join(ROOT, 'tests', 'golden', 'data')
    z = np.load(os.path.join(ROOT, 'tests', 'golden', case + '.npz'))
    GR0 = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d))
    sig = ic.sigma_levels(GR0, data_dir)
    np.testing.assert_allclose(sig, z['grid/sigma_vb'].ravel(), rtol=0, atol=1e-15)
    GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d), sigma_vb=z['grid/sigma_vb'].ravel())
    F = ModelFields(GR)
    ic.initialize_fields(GR, F, data_dir)
    for n, tol in [('HSURF', 1e-12), ('COLP', 1e-13), ('POTT', 1e-13), ('UWIND', 1e-14), ('VWIND', 1e-14)]:
        ref, got = z['ic/' + n], F.host[n]
        m = np.isfinite(ref)
        assert np.array_equal(m, np.isfinite(got)), n
        scale = np.max(np.abs(ref[m]))
        assert np.max(np.abs(got[m] - ref[m])) <= tol * scale, (n, np.max(np.abs(got[m] - ref[m])) / scale)


def _build_c_client(tmp_path):
    exe = str(tmp_path / 'c_client')
    r = subprocess.run(['gcc', '-O1', '-Wall', '-Werror', '-I' + os.path.join(ROOT, 'include'),
                        os.path.join(ROOT, 'examples', 'c_client.c'), '-o', exe, '-ldl', '-lm'],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_plain_c_client_builds_and_fails_loudly_without_gpu(tmp_path):
    """include/dycore.h is consumable from C (gcc -Wall -Werror) and every entry point the client uses
    resolves in the library; without a GPU dycore_create reports the error (exit code 3), with one the
    client runs to completion."""
    exe = _build_c_client(tmp_path)
    r = subprocess.run([exe, LIB], capture_output=True, text=True, timeout=300)
    assert r.returncode in (0, 3), r.stdout + r.stderr
    if r.returncode == 3:
        assert 'dycore_create' in r.stderr and 'device' in r.stderr.lower()


@pytest.mark.gpu
def test_plain_c_client_atmosphere_at_rest(tmp_path):
    """examples/c_client.c: a host in plain C drives the library; an atmosphere at rest stays at rest."""
    exe = _build_c_client(tmp_path)
    r = subprocess.run([exe, LIB], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'C_CLIENT_OK' in r.stdout, r.stdout + r.stderr


def test_bench_self_launch_command_and_no_gpu_touch(monkeypatch):
    """`python bench.py --gpus N` without a launcher: the parent builds the torch.distributed.run command
    for N ranks on 127.0.0.1, runs it as a child, relays stdout and returns its exit code -- and importing
    bench.py (all the parent does before that) loads neither torch nor the HIP library."""
    import subprocess
    import types
    r = subprocess.run([sys.executable, '-c',
                        'import sys; sys.argv=["bench.py"]; import bench; '
                        'print("torch" in sys.modules, any("dycore_hip" in (getattr(m, "__file__", "") or "") for m in sys.modules.values()))'],
                       cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.split() == ['False', 'False'], r.stdout + r.stderr
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        seen['cmd'], seen['env'] = cmd, env
        return types.SimpleNamespace(returncode=7, stdout='{"metric": "x"}\n')
    monkeypatch.setattr(subprocess, 'run', fake_run)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '2'])
    rc = bench.self_launch(types.SimpleNamespace(gpus=4))
    assert rc == 7
    cmd = seen['cmd']
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[-4:] == ['--gpus', '4', '--steps', '2'] and cmd[-5].endswith('bench.py')


def test_momentum_level_loop_has_no_compiler_inserted_drain():
    """The LDS-DMA momentum kernel counts its own DMA instructions; a compiler-inserted `s_waitcnt vmcnt` (or a spill)
    inside its level loop drains the levels in flight and costs ~10 % of the kernel without changing a result, so only
    the ISA shows it (profiles/r02_notes.md).  tools/check_mom_isa.py compiles the device code to assembly (no GPU) and
    returns 1 on a finding in a production instance."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_mom_isa.py')], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if 'fw=1 turb=0' in l]
    assert len(lines) >= 6 and all('findings: none' in l for l in lines), r.stdout


def test_term_switch_tables_agree():
    """The per-term switches of namelist.py:62-88 travel as one bit mask: the bits in include/dycore.h (DYCORE_TERM_*), in
    oracle/dycore_oracle.h (ORACLE_TERM_*), in climate_model_amd.namelist.TERM_BITS and in the tests' own table are the same,
    the ctypes mirror of DycoreConfig has the header's layout, and Namelist.term_off sets exactly the bits of the entries at 0."""
    import ctypes as C
    from climate_model_amd.namelist import TERM_BITS, Namelist
    from climate_model_amd.hip_backend import DycoreConfig
    from golden_util import TERM_BITS as TEST_BITS
    assert TERM_BITS == TEST_BITS
    hdr = open(os.path.join(ROOT, 'include', 'dycore.h')).read()
    c_bits = {m.group(1): 1 << int(m.group(2)) for m in re.finditer(r'#define DYCORE_TERM_(\w+)\s+\(1 << (\d+)\)', hdr)}
    ohdr = open(os.path.join(ROOT, 'oracle', 'dycore_oracle.h')).read()
    o_bits = {m.group(1): 1 << int(m.group(2)) for m in re.finditer(r'ORACLE_TERM_(\w+) = 1 << (\d+)', ohdr)}
    assert c_bits == o_bits and len(c_bits) == len(TERM_BITS) == 14
    for name, bit in TERM_BITS.items():         # i_UVFLX_hor_adv -> UVFLX_HOR_ADV, i_COLP_main_switch -> COLP_MAIN
        key = name[2:].upper().replace('_MAIN_SWITCH', '_MAIN')
        assert c_bits[key] == bit, name
    assert Namelist().term_off == 0
    assert Namelist(i_UVFLX_coriolis=0, i_moist_num_dif=0).term_off == TERM_BITS['i_UVFLX_coriolis'] | TERM_BITS['i_moist_num_dif']
    # struct layout: 11 int32 + term_off + 4 reserved int32 (= 16 x 4 bytes), then the double and the pointer
    names = [f[0] for f in DycoreConfig._fields_]
    assert names.index('term_off') == names.index('i_moist_clip') + 1 and DycoreConfig.term_off.offset == 44
    assert DycoreConfig.pair_top.offset == 64 and C.sizeof(DycoreConfig) == 80
