"""Latitude-band logic and the torch.distributed exchange path on CPU (gloo, 2 ranks)."""
import os
import socket

import numpy as np
import pytest

from climate_model_amd.bands import (BandLayout, BandHalo, SEND_NORTH, SEND_SOUTH,
                                     RECV_FROM_SOUTH, RECV_FROM_NORTH, gather_global)
from climate_model_amd.fields import FIELD_TABLE
from band_util import NumpyBandCore, global_test_fields

EXCHANGED = ['UFLX', 'VFLX', 'WWIND', 'COLP_NEW', 'UWIND', 'VWIND', 'POTT']


def test_layout_partitions_rows():
    for ny, nb in [(45, 2), (45, 4), (720, 8), (18, 3), (16, 8)]:
        L = BandLayout(ny, nb)
        assert sum(L.heights) == ny and min(L.heights) >= 2
        assert L.offsets[0] == 0 and all(L.offsets[r + 1] == L.offsets[r] + L.heights[r] for r in range(nb - 1))
        assert L.south(0) is None and L.north(nb - 1) is None
    with pytest.raises(ValueError):
        BandLayout(7, 4)


def test_plan_is_symmetric():
    for ph in ('A', 'B'):
        assert len(SEND_NORTH[ph]) == len(RECV_FROM_SOUTH[ph])
        assert len(SEND_SOUTH[ph]) == len(RECV_FROM_NORTH[ph])


def test_slice_and_gather_roundtrip():
    nx, ny, nz = 12, 17, 3
    G = global_test_fields(nx, ny, nz, EXCHANGED)
    for nb in (1, 2, 3, 5):
        L = BandLayout(ny, nb)
        for name in EXCHANGED:
            stgy = FIELD_TABLE[name][1]
            bands = [L.slice_rows(G[name], r, stgy) for r in range(nb)]
            np.testing.assert_array_equal(gather_global(L, bands, name), G[name])


def _exchange_worker(rank, world, port, nx, ny, nz, q):
    import torch
    import torch.distributed as dist
    from climate_model_amd.bands import DistributedBands
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        L = BandLayout(ny, world)
        G = global_test_fields(nx, ny, nz, EXCHANGED)
        local = {n: L.slice_rows(G[n], rank, FIELD_TABLE[n][1]) for n in EXCHANGED}
        nyl = L.ny(rank)
        # wipe the halo rows a neighbour must provide
        for n, a in local.items():
            stgy = FIELD_TABLE[n][1]
            if L.south(rank) is not None:
                a[:, 0] = -1
            if L.north(rank) is not None:
                a[:, nyl + 1:] = -1
        core = NumpyBandCore(nx, nyl, nz, local)
        bands = DistributedBands(core, L, rank, 'cpu')
        bands.connect()             # peer connections first (bench.py): touches no field
        bands.exchange('A')
        bands.exchange('B')
        bad = []
        for n, a in local.items():
            stgy = FIELD_TABLE[n][1]
            want = L.slice_rows(G[n], rank, stgy)
            rows = [0, nyl + 1] + ([nyl + 2] if n == 'VFLX' else [])
            for r in rows:
                if not np.array_equal(a[:, r], want[:, r]):
                    bad.append((n, r))
        q.put((rank, bad))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_gloo_neighbour_exchange(world):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, 10, 13, 3, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, bad in res:
        assert bad == [], 'rank %d: wrong halo rows %s' % (rank, bad)


# ---- first contact of the in-library RCCL transport: every rank takes the same branch --------------------------
class _FakeNative:
    closed = False

    def connect(self):
        pass

    def close(self):
        _FakeNative.closed = True


class _FakeCore:
    dtype = 'float64'

    def sync(self):
        pass


def _negotiate_worker(rank, world, port, case, q):
    import torch.distributed as dist
    from climate_model_amd import bands, hip_backend
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        def fake_id(dtype=None):
            if case == 'no_rccl' and rank == 1:
                raise hip_backend.DycoreError('librccl.so could not be loaded')
            return bytes(128)

        def fake_probe(*a, **k):
            return (False, 'probe failed (exit 1)') if (case == 'probe' and rank == 0) else (True, None)

        def fake_rccl(core, uid, r, n, south, north):
            if case == 'init' and rank == 1:
                raise hip_backend.DycoreError('ncclCommInitRank failed: unhandled system error')
            return _FakeNative()

        hip_backend.bands_unique_id = fake_id
        bands._probe_native = fake_probe
        hip_backend.NativeBands.rccl = staticmethod(fake_rccl)
        nb, why = bands._negotiate_native(_FakeCore(), BandLayout(12, world), rank, 'cpu', None)
        q.put((rank, nb is not None, why, _FakeNative.closed))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('case', ['no_rccl', 'probe', 'init', 'ok'])
def test_native_transport_is_all_or_nothing(case):
    """bands._negotiate_native on 2 gloo ranks with the library calls replaced: a failure on ONE rank (librccl missing,
    probe child failed, ncclCommInitRank error) sends BOTH to the torch.distributed transport, and a handle that was
    already created on the healthy rank is closed again."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_negotiate_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=120) for _ in procs))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if case == 'ok':
        assert res[0][0] and res[1][0] and res[0][1] is None
    else:
        assert not res[0][0] and not res[1][0], res
        assert res[0][1] and res[1][1]          # both know why
        if case == 'init':
            assert res[0][2]                    # rank 0's handle was closed again


def test_watchdog_turns_a_hang_into_exit_86():
    import subprocess
    import sys
    code = ("import time\nfrom climate_model_amd.bands import _Watchdog\n"
            "with _Watchdog(0.5, 'the test call'):\n    time.sleep(30)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=60)
    assert r.returncode == 86 and 'did not return within' in r.stderr
    code = "from climate_model_amd.bands import _Watchdog\nwith _Watchdog(30, 'x'):\n    pass\nprint('fine')\n"
    r = subprocess.run([sys.executable, '-c', code], cwd=root, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and 'fine' in r.stdout
