"""Host-side cost of one neighbour exchange through torch.distributed (nccl = RCCL): 4 P2POps to self
on a side stream, 1 MB each, issue time per batch."""
import os, sys, time
import torch
import torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29655')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
comm = torch.cuda.Stream()
s = torch.zeros(2 * 131072, dtype=torch.float64, device='cuda'); r = torch.empty_like(s)
def batch():
    ops = [dist.P2POp(dist.isend, s[:131072], 0), dist.P2POp(dist.irecv, r[:131072], 0),
           dist.P2POp(dist.isend, s[131072:], 0), dist.P2POp(dist.irecv, r[131072:], 0)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
with torch.cuda.stream(comm):
    for _ in range(5): batch()
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
with torch.cuda.stream(comm):
    for _ in range(n): batch()
t_issue = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
print('batch_isend_irecv of 4 ops: host issue %.1f us, wall %.1f us per batch' % (t_issue * 1e6, t_all * 1e6))
dist.destroy_process_group()
