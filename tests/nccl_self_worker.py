"""RCCL transport smoke on ONE GPU: a middle band sends its packed rows to itself (world_size 1,
send/recv to rank 0 inside one group), on the comm stream of the overlapped schedule, and unpacks
them -- its north halo rows must then hold its own southern rows and vice versa.  Two ranks cannot
share a GPU under RCCL ("Duplicate GPU detected"), so this is as much of the nccl path as one GPU
can exercise: process group, P2POp batches on a side stream, stream/event ordering against the
pack and unpack kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

from climate_model_amd import HipDycore
from climate_model_amd.bands import BandLayout, BandHalo, _TorchBuffers, _Streams, _merged
from climate_model_amd.fields import FIELD_TABLE, field_shape


def main():
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    dev = torch.device('cuda', 0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    nx, nyg, nz = 96, 30, 8
    L = BandLayout(nyg, 3)
    ny = L.ny(1)
    core = HipDycore(nx, ny, nz, ny_global=nyg, j_offset=L.j_offset(1), stream=stream.cuda_stream)
    rng = np.random.default_rng(0)
    halo0 = BandHalo(core, ny, True, True)
    names = sorted({n for ph in ('A', 'B') for k in ('send_south', 'send_north') for n in halo0.resolved(ph, 1)[k][0]})
    host = {}
    for n in names:
        host[n] = rng.standard_normal(field_shape(n, nx, ny, nz))
        core.upload(n, host[n])
    halo = BandHalo(core, ny, True, True)
    bufs = _TorchBuffers(halo, dev)
    streams = _Streams([core], dev)
    for rep in range(3):                    # several rounds: buffer reuse across exchanges
        for ph in ('A', 'B'):
            core.pack_rows(*_merged(halo, ph, 1, 'send'), bufs.send[ph].data_ptr())
            streams.comm.wait_stream(streams.main)
            with torch.cuda.stream(streams.comm):
                core.use_stream(streams.comm.cuda_stream)
                (ss, sn), (rs, rn) = bufs.split_send[ph], bufs.split_recv[ph]
                assert ss == rn and sn == rs
                s, r = bufs.send[ph], bufs.recv[ph]
                ops = [dist.P2POp(dist.isend, s[:ss], 0), dist.P2POp(dist.irecv, r[rs:rs + rn], 0),
                       dist.P2POp(dist.isend, s[ss:ss + sn], 0), dist.P2POp(dist.irecv, r[:rs], 0)]
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
                core.unpack_rows(*_merged(halo, ph, 1, 'recv'), bufs.recv[ph].data_ptr())
                core.use_stream(None)
                streams.done[ph].record(streams.comm)
            streams.pending[ph] = True
            streams.wait(ph)
    core.sync()
    # what went south came back as the north halo rows, what went north as the south halo row: walk
    # the plan (send list of one direction = receive list of the opposite one)
    got = {n: core.download(n) for n in names}
    checked = 0
    for ph in ('A', 'B'):
        plan = halo.resolved(ph, 1)
        for send, recv in (('send_south', 'recv_north'), ('send_north', 'recv_south')):
            for n, srow, rrow in zip(plan[send][0], plan[send][1], plan[recv][1]):
                assert plan[recv][0] == plan[send][0]
                want = host[n][:, srow].copy()          # a packed row carries the periodic images of its interior columns
                want[0], want[nx + 1] = want[nx], want[1]
                if want.shape[0] > nx + 2:
                    want[nx + 2] = want[2]
                np.testing.assert_array_equal(got[n][:, rrow], want, err_msg='%s row %d -> %d' % (n, srow, rrow))
                checked += 1
    assert checked >= 14
    for n in names:
        np.testing.assert_array_equal(got[n][:, 1:ny + 1], host[n][:, 1:ny + 1], err_msg=n + ' owned rows untouched')
    core.close()
    dist.destroy_process_group()
    print('NCCL_SELF_OK')


if __name__ == '__main__':
    main()
