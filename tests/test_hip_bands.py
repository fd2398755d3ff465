"""Band-decomposed HIP path on one GPU: N bands must reproduce the 1-band result bit for
bit (SURVEY.md 8e parity check).  In-process bands exchange by device copies; the
2-process variant goes through torch.distributed (gloo, host-staged) like the N>1 bench."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from golden_util import load_golden
from hip_util import grid_from_golden

from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
from climate_model_amd.bands import BandLayout, LocalBands, band_grid_arrays, gather_global
from climate_model_amd.fields import FIELD_TABLE
from climate_model_amd.synthetic import initialize_fields

pytestmark = pytest.mark.gpu
CHECK = ['COLP', 'UWIND', 'VWIND', 'POTT', 'PHI', 'PHIVB', 'PVTF', 'PVTFVB', 'POTTVB', 'WWIND', 'UFLX', 'VFLX']
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _initial_state(d, nz, i_turb, i_moist=0):
    GR = Grid(Namelist(nz=nz, dlat_deg=d, dlon_deg=d, i_UVFLX_vert_turb=i_turb, i_POTT_vert_turb=i_turb,
                       i_moist_main_switch=i_moist))
    F = ModelFields(GR)
    initialize_fields(GR, F)
    if i_turb:
        rng = np.random.default_rng(9)
        for n, lo, hi in [('KMOM', 0.5, 20.), ('KHEAT', 0.5, 20.), ('SSHFLX', -50., 150.),
                          ('SMOMXFLX', -0.2, 0.2), ('SMOMYFLX', -0.2, 0.2)]:
            F.host[n][...] = rng.uniform(lo, hi, F.host[n].shape)
    names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB', 'KMOM', 'KHEAT',
             'SSHFLX', 'SMOMXFLX', 'SMOMYFLX', 'dPOTTdt_RAD']
    if i_moist:
        from climate_model_amd.synthetic import initialize_moisture
        initialize_moisture(GR, F)
        names += ['QV', 'QC', 'SLHFLX']
    return GR, {n: F.host[n] for n in names}


def _single(GR, state, i_turb, nsteps, i_moist=0, check=None):
    core = HipDycore(GR.nx, GR.ny, GR.nz, i_turb=i_turb, i_moist=i_moist)
    core.set_grid(GR.GRF['HIP'])
    for n, a in state.items():
        core.upload(n, a)
    core.primary_diag()
    core.step_matsuno(GR.dt, nsteps)
    out = {n: core.download(n) for n in (check or CHECK)}
    core.close()
    return out


@pytest.mark.parametrize('overlap', [False, True, 'native'])
@pytest.mark.parametrize('nbands', [2, 3, 5])
@pytest.mark.parametrize('i_turb,i_moist', [(0, 0), (1, 0), (0, 1), (1, 1)])
def test_local_bands_bitwise_equal_single(nbands, i_turb, i_moist, overlap):
    """overlap=True: the exchanges run on a second stream behind the inner rows (the schedule of
    the multi-GPU run); overlap=False: plain part / exchange sequence with host synchronisation;
    'native': the same overlapped schedule issued by the library itself (dycore_bands_step_matsuno)."""
    _local_bands_vs_single(nbands, i_turb, i_moist, overlap)


@pytest.mark.parametrize('overlap', [False, True, 'native'])
@pytest.mark.parametrize('nbands,i_moist', [(2, 0), (3, 1), (5, 0)])
def test_local_bands_on_the_whole_substep_kernel(nbands, i_moist, overlap, monkeypatch):
    """DYCORE_BANDS_STEP_FUSED=1: a band's momentum and temperature kernels are one launch too (k_momentum_dma<..., WITH_TEMP>
    in pieces: inner tile rows in two halves around the edge rows; by default only bands of six rounds of tiles and more --
    slower on small bands, profiles/r03_notes.md).  Same bits as the single domain; the handles really launch that kernel."""
    monkeypatch.setenv('DYCORE_BANDS_STEP_FUSED', '1')
    stats = _local_bands_vs_single(nbands, 0, i_moist, overlap, want_stats=True)
    assert all(st.get('k_substep_fused', (0, 0))[0] > 0 and st.get('k_temp_canon', (0, 0))[0] == 0 for st in stats), stats


def _local_bands_vs_single(nbands, i_turb, i_moist, overlap, want_stats=False):
    native = overlap == 'native'
    overlap = bool(overlap)
    import torch
    stream = torch.cuda.Stream()
    GR, state = _initial_state(4, 8, i_turb, i_moist)
    check = CHECK + (['QV', 'QC'] if i_moist else [])
    ref = _single(GR, state, i_turb, 3, i_moist, check)
    L = BandLayout(GR.ny, nbands)
    cores = []
    for r in range(nbands):
        c = HipDycore(GR.nx, L.ny(r), GR.nz, i_turb=i_turb, ny_global=GR.ny, j_offset=L.j_offset(r), i_moist=i_moist,
                      stream=stream.cuda_stream)
        c.set_grid(band_grid_arrays(GR.GRF['HIP'], L, r))
        for n, a in state.items():
            c.upload(n, L.slice_rows(a, r, FIELD_TABLE[n][1]))
        c.primary_diag()
        cores.append(c)
    lb = LocalBands(cores, L, torch.device('cuda:0'), overlap=overlap, native=native)
    assert lb.overlap == overlap        # the split parts exist for the turbulence-coupled fused step too
    assert (lb.native is not None) == native
    lb.step_matsuno(GR.dt, 2)
    if want_stats:
        for c in cores:
            c.profile(True)
    lb.step_matsuno(GR.dt, 1)          # a second call continues from halo rows the first one left in place
    stats = [c.kernel_stats() for c in cores] if want_stats else None
    for n in check:
        got = gather_global(L, [c.download(n) for c in cores], n)
        np.testing.assert_array_equal(got, ref[n], err_msg='%s incl. halos (%d bands)' % (n, nbands))
    for c in cores:
        c.close()
    return stats


@pytest.mark.parametrize('nproc', [2, 4])
def test_multi_process_gloo_bands_equal_single(tmp_path, nproc):
    """bench.py-style launch: the ranks share the one GPU and exchange through torch.distributed (gloo with
    host staging); with 4 ranks two bands have a neighbour on both sides."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = tmp_path / 'bands.npz'
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', DYCORE_BANDS_BACKEND='gloo', DYCORE_BANDS_ONE_GPU='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'tests', 'band_worker.py'), str(out)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    GR, state = _initial_state(4, 8, 0)
    ref = _single(GR, state, 0, 3)
    z = np.load(out)
    for n in ('COLP', 'UWIND', 'VWIND', 'POTT'):
        np.testing.assert_array_equal(z[n], ref[n], err_msg=n)


def _n_devices():
    import torch
    return torch.cuda.device_count()        # (counting devices does not initialise the GPU)


@pytest.mark.parametrize('nproc,probe', [(2, '1'), (2, '0'), (4, '1')])
def test_rccl_bands_between_distinct_devices(tmp_path, nproc, probe):
    """The multi-GPU path proper, run wherever the box has the devices: one rank per GPU, process group on nccl (= RCCL),
    DistributedBands brings up the in-library transport (probe child first, then ncclCommInitRank next to torch's own
    communicator), three Matsuno steps through ncclSend / ncclRecv between real peers -- bitwise equal to the
    single-domain run.  Skipped on a one-GPU box (RCCL refuses two ranks on one device); there the same schedule is
    covered by the self-neighbour RCCL test above and the gloo runs."""
    if _n_devices() < nproc:
        pytest.skip('needs %d GPUs, %d visible' % (nproc, _n_devices()))
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    out = tmp_path / 'bands.npz'
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', DYCORE_BANDS_BACKEND='nccl', DYCORE_EXPECT_NATIVE='1',
               DYCORE_BANDS_PROBE=probe)
    env.pop('DYCORE_BANDS_ONE_GPU', None)
    env.pop('DYCORE_BANDS_NATIVE', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'tests', 'band_worker.py'), str(out)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    GR, state = _initial_state(4, 8, 0)
    ref = _single(GR, state, 0, 3)
    z = np.load(out)
    for n in ('COLP', 'UWIND', 'VWIND', 'POTT'):
        np.testing.assert_array_equal(z[n], ref[n], err_msg=n)


def test_bands_probe_child_on_one_gpu():
    """python -m climate_model_amd.bands_probe, the child every rank starts before it commits to the in-library
    transport: here as a single rank that is its own neighbour on both sides (two ranks cannot share a GPU under RCCL)."""
    from climate_model_amd.hip_backend import bands_unique_id
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
    cmd = [sys.executable, '-m', 'climate_model_amd.bands_probe', bands_unique_id().hex(), '0', '1', '0', '0', '0', 'float64']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'BANDS_PROBE_OK' in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_bench_two_ranks_rehearsal():
    """The bare command the round driver runs, `python bench.py --gpus 2 ...` with no launcher and no
    WORLD_SIZE: bench.py starts torch.distributed.run as a child before it touches the GPU, relays the
    JSON line and returns the child's exit code.  Two ranks on the one GPU, gloo instead of RCCL, small grid."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', DYCORE_BANDS_BACKEND='gloo', DYCORE_BANDS_ONE_GPU='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--dlat', '2', '--nz', '16']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-3000:]
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 3 and j['value'] > 0 and j['scaling'] == 'strong'
    assert j['config']['parallelism'] == 'latitude bands x2'
    roof = j['roofline']
    assert roof and 0 < roof['frac'] <= 1 and 0 < roof['step_frac'] <= 1
    assert all(0 < k['frac'] <= 1 for k in roof['kernels'].values())


def test_bench_self_launch_propagates_failure():
    """A failing child (here: more bands than the grid has row pairs) must fail the bare command too."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', DYCORE_BANDS_BACKEND='gloo', DYCORE_BANDS_ONE_GPU='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--dlat', '90', '--nz', '4']
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0


@pytest.mark.parametrize('i_moist,prec', [(0, 'f64'), (1, 'f64'), (0, 'f32'), (1, 'f32')])
def test_native_band_step_over_rccl(i_moist, prec):
    """dycore_bands_create_rccl / dycore_bands_step_matsuno: ncclSend / ncclRecv issued by the library on its comm
    stream (tests/native_rccl_worker.py: a band that is its own neighbour, against the device-copy transport).
    'f32': libdycore_hip_f32.so -- message counts are in elements of the build's `real`, the RCCL datatype must follow."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'native_rccl_worker.py'), str(i_moist), prec], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'NATIVE_RCCL_OK' in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_rccl_transport_self_exchange():
    """nccl (= RCCL) send/recv of the packed band buffers on the comm stream, one rank sending to itself
    (tests/nccl_self_worker.py)."""
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
               RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'nccl_self_worker.py')], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'NCCL_SELF_OK' in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
