"""Latitude-band decomposition of the lat-lon grid across the GPUs of one node.

New functionality (the reference is single-device; its "halo exchange" is the in-array
copy of misc_boundaries.py).  The grid is cut into contiguous bands of latitude rows, one
band per GPU / process; longitude stays whole, so the periodic-i boundary remains local.
A band keeps the reference's halo convention (one row north and south; two for the
y-staggered VFLX) and refreshes those rows from its neighbours twice per Matsuno substep:

  phase 'A' after continuity : UFLX, VFLX, WWIND, COLP_NEW   (feed the 3x3 momentum fluxes)
  phase 'B' after Euler step : UWIND, VWIND, POTT            (the new state)

The overlapped schedule (_overlapped_step) sends phase B in two pieces: 'B1' = UWIND, VWIND as
soon as the momentum kernel is done, and the POTT (QV, QC) rows together with phase A of the NEXT
substep ('AB2'), so that every message travels behind a kernel that does not need it and a
substep costs two exchanges, not three ('B2' alone is the flush at the end of a call).

Ownership: band r owns mass rows j in [j0, j0+ny) and the v-rows (js) on the southern edge
of each of them; the northernmost band also owns js = ny_global+1 (the pole).  Every
exchanged row therefore has exactly one owner, and an N-band run reproduces the 1-band
result bit for bit (same arithmetic per cell).  The polar bands apply the wall rule of
misc_boundaries.py:36-41 instead of a receive.

Transport: neighbour send/recv through torch.distributed -- backend "nccl" (= RCCL over
xGMI on MI355X) on device buffers, or "gloo" with host staging for tests.  There is no
collective on the path: each neighbour pair uses its own point-to-point link.
"""
import numpy as np

from .fields import FIELD_TABLE

# (field, row to send); 'N' stands for the band's last owned row ny
SEND_NORTH = {
    'A': [('UFLX', 'N'), ('VFLX', 'N'), ('WWIND', 'N'), ('COLP_NEW', 'N')],
    'B': [('UWIND', 'N'), ('VWIND', 'N'), ('POTT', 'N')],
    'B1': [('UWIND', 'N'), ('VWIND', 'N')],
    'B2': [('POTT', 'N')],
    'AB2': [('UFLX', 'N'), ('VFLX', 'N'), ('WWIND', 'N'), ('COLP_NEW', 'N'), ('POTT', 'N')],
}
SEND_SOUTH = {
    'A': [('UFLX', 1), ('VFLX', 1), ('VFLX', 2), ('WWIND', 1), ('COLP_NEW', 1)],
    'B': [('UWIND', 1), ('VWIND', 1), ('POTT', 1)],
    'B1': [('UWIND', 1), ('VWIND', 1)],
    'B2': [('POTT', 1)],
    'AB2': [('UFLX', 1), ('VFLX', 1), ('VFLX', 2), ('WWIND', 1), ('COLP_NEW', 1), ('POTT', 1)],
}
PHASES = ('A', 'B', 'B1', 'B2', 'AB2')
STATE_FIELDS = ('UWIND', 'VWIND', 'POTT', 'QV', 'QC')
# rows at which the RECEIVER stores what its southern / northern neighbour sent
# (offsets relative to the receiver's own ny: 'N+1' = north halo row, 'N+2' = second one)
RECV_FROM_SOUTH = {'A': [0, 0, 0, 0], 'B': [0, 0, 0], 'B1': [0, 0], 'B2': [0], 'AB2': [0, 0, 0, 0, 0]}
RECV_FROM_NORTH = {'A': ['N+1', 'N+1', 'N+2', 'N+1', 'N+1'], 'B': ['N+1', 'N+1', 'N+1'], 'B1': ['N+1', 'N+1'], 'B2': ['N+1'],
                   'AB2': ['N+1', 'N+1', 'N+2', 'N+1', 'N+1', 'N+1']}


class BandLayout:
    """Split of ny_global latitude rows into nbands contiguous bands (pure logic)."""

    def __init__(self, ny_global, nbands):
        if nbands < 1 or ny_global < 2 * nbands:
            raise ValueError('need at least 2 rows per band: ny=%d, bands=%d' % (ny_global, nbands))
        self.ny_global, self.nbands = int(ny_global), int(nbands)
        base, rem = divmod(self.ny_global, self.nbands)
        self.heights = [base + (1 if r < rem else 0) for r in range(self.nbands)]
        self.offsets = [sum(self.heights[:r]) for r in range(self.nbands)]   # j_offset of band r

    def ny(self, r):
        return self.heights[r]

    def j_offset(self, r):
        return self.offsets[r]

    def south(self, r):
        return r - 1 if r > 0 else None

    def north(self, r):
        return r + 1 if r < self.nbands - 1 else None

    def slice_rows(self, arr, r, stgy):
        """Band r's local array (halo rows included) out of a global [i, j, k] array."""
        j0 = self.offsets[r]
        return np.ascontiguousarray(arr[:, j0:j0 + self.heights[r] + 2 + stgy])

    def owned_rows(self, r, stgy):
        """Local row range [lo, hi) a band owns (used to reassemble a global field)."""
        hi = self.heights[r] + 1
        if stgy and r == self.nbands - 1:
            hi += 1
        return 1, hi


def _resolve(rows, ny):
    out = []
    for r in rows:
        if r == 'N':
            out.append(ny)
        elif r == 'N+1':
            out.append(ny + 1)
        elif r == 'N+2':
            out.append(ny + 2)
        else:
            out.append(int(r))
    return out


class BandHalo:
    """Message plan of one band: what to pack for / unpack from each neighbour."""

    def __init__(self, core, ny, has_south, has_north):
        self.core, self.ny = core, ny
        self.has_south, self.has_north = has_south, has_north
        self._resolved = {}
        self.plan = {}
        moist = bool(getattr(core, 'i_moist', False))       # QV, QC travel with the state (phase B)
        # a core whose fused step re-forms the mass fluxes from the winds (flux_from_winds) needs the
        # 2-D column pressures at the u- / v-points where the other needs UFLX, VFLX (same rows), and
        # the second v-row beyond its northern edge with the winds
        fw = bool(getattr(core, 'flux_from_winds', False))
        swap = {'UFLX': 'UCOL', 'VFLX': 'VCOL'} if fw else {}
        # ... and with the scalar state the diagnostics of that state as their OWNER computed them (PVTFVB, and PHIVB / PHI /
        # POTTVB / PVTF depending on what the handle's step stores): a band does not re-diagnose rows it receives
        diag = [(n, None) for n in getattr(core, 'band_diag_fields', ())]
        for ph in PHASES:
            extra = ([('QV', None), ('QC', None)] if moist else []) + diag if ph in ('B', 'B2', 'AB2') else []
            north = [(swap.get(f, f), r) for f, r in SEND_NORTH[ph]]
            south = [(swap.get(f, f), r) for f, r in SEND_SOUTH[ph]]
            rs, rn = list(RECV_FROM_SOUTH[ph]), list(RECV_FROM_NORTH[ph])
            if fw and ph in ('B', 'B1'):
                south.append(('VWIND', 2))
                rn.append('N+2')
            sn = [f for f, _ in north] + [f for f, _ in extra]
            ss = [f for f, _ in south] + [f for f, _ in extra]
            self.plan[ph] = dict(
                send_north=(sn, _resolve([r for _, r in north] + ['N'] * len(extra), ny)),
                send_south=(ss, _resolve([r for _, r in south] + [1] * len(extra), ny)),
                recv_south=(sn, _resolve(rs + [0] * len(extra), ny)),      # neighbour's send_north list
                recv_north=(ss, _resolve(rn + ['N+1'] * len(extra), ny)),  # neighbour's send_south list
            )

    def sizes(self, ph):
        p = self.plan[ph]
        return {k: self.core.rows_size(p[k][0]) for k in p}

    def resolved(self, ph, sub):
        """The plan with state-field names mapped to the buffer that holds them (the fused library
        step keeps the predictor state in *_EST scratch fields): for 'B*' the state substep `sub`
        wrote, for 'AB2' the state substep `sub` evaluates on = what the previous substep wrote."""
        key = (ph, sub)
        if key not in self._resolved:
            fn = getattr(self.core, 'substep_field', None)
            src = 1 - sub if ph == 'AB2' else sub
            out = {}
            for k, (names, rows) in self.plan[ph].items():
                out[k] = ([fn(src, n) if (fn and n in STATE_FIELDS) else n for n in names], rows)
            self._resolved[key] = out
        return self._resolved[key]


class _TorchBuffers:
    """One send and one receive buffer per phase, laid out [south part | north part], so that a
    band packs (and unpacks) both directions with ONE kernel launch."""

    def __init__(self, halo, device):
        import torch
        self.send, self.recv, self.split_send, self.split_recv = {}, {}, {}, {}
        # message elements have the working precision of the handle (rows_size counts elements)
        dt = torch.float32 if str(getattr(halo.core, 'dtype', 'float64')) == 'float32' else torch.float64
        for ph in PHASES:
            sz = halo.sizes(ph)
            ss = sz['send_south'] if halo.has_south else 0
            sn = sz['send_north'] if halo.has_north else 0
            rs = sz['recv_south'] if halo.has_south else 0
            rn = sz['recv_north'] if halo.has_north else 0
            self.send[ph] = torch.empty(max(ss + sn, 1), dtype=dt, device=device)
            self.recv[ph] = torch.empty(max(rs + rn, 1), dtype=dt, device=device)
            self.split_send[ph] = (ss, sn)
            self.split_recv[ph] = (rs, rn)


def _merged(halo, ph, sub, kind):
    """Concatenated (names, rows) of the south and north parts of a pack ('send') or unpack
    ('recv') list, matching the [south | north] buffer layout."""
    p = halo.resolved(ph, sub)
    names, rows = [], []
    if halo.has_south:
        names += p[kind + '_south'][0]; rows += p[kind + '_south'][1]
    if halo.has_north:
        names += p[kind + '_north'][0]; rows += p[kind + '_north'][1]
    return names, rows


class _Streams:
    """The two HIP streams of an overlapped band step: `main` is the stream the core launches on,
    `comm` carries pack -> transport -> unpack (+ the diagnostics of the received rows).  One
    event per exchange phase marks "halo rows of this phase are in place"."""

    def __init__(self, cores, device):
        import torch
        self.torch = torch
        ptrs = {c.stream_ptr for c in cores}
        if len(ptrs) != 1:
            raise ValueError('all bands of one process must launch on the same stream')
        ptr = ptrs.pop()
        self.main = torch.cuda.ExternalStream(ptr, device=device) if ptr else torch.cuda.default_stream(device)
        self.comm = torch.cuda.Stream(device=device)
        self.done = {ph: torch.cuda.Event() for ph in PHASES}
        self.pending = {ph: False for ph in PHASES}

    def wait(self, ph):
        """main stream: do not pass this point before exchange `ph` has delivered."""
        if self.pending[ph]:
            self.main.wait_event(self.done[ph])
            self.pending[ph] = False


def _overlapped_step(cores, streams, deliver, dt, nsteps, with_rho):
    """One schedule for one band per process (DistributedBands) and for several bands in one
    process (LocalBands).  Per substep, on the main stream (| = point where it waits):

        |B1   continuity                          pack AB2 (fluxes' factors, WWIND, COLP_NEW + POTT rows
              momentum, inner tiles                         of the state this substep evaluates on)
        |AB2  momentum, edge tiles                pack B1  (new winds)
              temperature (+ moisture), halos

    and on the comm stream, behind each pack: transport, unpack.  AB2 travels behind the inner momentum tiles, B1 behind the temperature
    kernel and the continuity kernel of the next substep's start.  The POTT rows written by the
    last substep of a call go out alone ('B2') before it returns.
    deliver(ph, sub) issues transport + unpack on the current stream."""
    torch = streams.torch
    C0 = cores[0]

    def on_comm(ph, sub):
        for c in cores:
            c.pack_rows(*_merged(c._band_halo, ph, sub, 'send'), c._band_bufs.send[ph].data_ptr())
        streams.comm.wait_stream(streams.main)
        with torch.cuda.stream(streams.comm):
            for c in cores:
                c.use_stream(streams.comm.cuda_stream)
            try:
                deliver(ph, sub)
            finally:
                for c in cores:
                    c.use_stream(None)
            streams.done[ph].record(streams.comm)
        streams.pending[ph] = True

    # a turbulence-coupled step re-diagnoses RHO / RHOVB of the whole band, halo rows included, when it begins: the POTT
    # rows of the state it starts from must be in place then, so they go out alone at the end of every step
    turb = bool(getattr(C0, 'i_turb', False))
    for step in range(nsteps):
        if turb:
            streams.wait('B2')
        for c in cores:
            c.step_begin(with_rho)
        for sub in (0, 1):
            streams.wait('B1')
            streams.wait('B2')
            for c in cores:
                c.substep_part(dt, sub, 0)
            on_comm('AB2', sub)
            # (the library's own schedule, csrc/dycore_bands.h: a handle on the whole-substep kernel hides B1 behind the second
            # half of its inner tile rows; on two kernels INNER_A is the whole inner piece, INNER_B nothing, and part TEMP --
            # the temperature kernel -- hides it)
            for c in cores:
                c.substep_part(dt, sub, C0.PART_MOM_INNER_A)
            streams.wait('AB2')
            for c in cores:
                c.substep_part(dt, sub, C0.PART_MOM_EDGE)
            on_comm('B1', sub)
            for c in cores:
                c.substep_part(dt, sub, C0.PART_MOM_INNER_B)
            for c in cores:
                c.substep_part(dt, sub, C0.PART_TEMP)
        if turb and step < nsteps - 1:
            on_comm('B2', 1)
    on_comm('B2', 1)        # leave with the halo rows of the final state in place and every stream joined
    streams.wait('B1')
    streams.wait('B2')


def _all_agree(dist, group, device, ok):
    """True when `ok` holds on every rank of the group (one all_reduce(MIN) on the group's own backend)."""
    import torch
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if dist.get_backend(group) == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(int(t.item()))


class _Watchdog:
    """Bounds a call that may block for ever inside a communication library (ncclCommInitRank and the first
    send / recv of a connection have no timeout of their own): if the block is still running after `seconds`, the
    process prints why and exits with code 86 -- a failed run the launcher sees, not a hung one.  Nothing is
    re-executed: a fresh attempt is a fresh child of whoever launched this process (bench.py's own launcher retries
    once with DYCORE_BANDS_NATIVE=0)."""

    def __init__(self, seconds, what):
        import threading
        self.what, self.seconds = what, seconds
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True

    def _fire(self):
        import os
        import sys
        sys.stderr.write('climate_model_amd.bands: %s did not return within %g s; giving up (exit 86). '
                         'DYCORE_BANDS_NATIVE=0 selects the torch.distributed transport.\n' % (self.what, self.seconds))
        sys.stderr.flush()
        os._exit(86)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


def _probe_native(unique_id, rank, nranks, south, north, device_index, dtype, timeout):
    """First contact of the in-library RCCL transport between THESE peers, in a child process: a small band per rank,
    communicator, connect, two steps (python -m climate_model_amd.bands_probe).  A child that fails, or is still
    running after `timeout` seconds and is killed, costs this process nothing: it reports False and every rank then
    agrees on the torch.distributed transport."""
    import os
    import subprocess
    import sys
    cmd = [sys.executable, '-m', 'climate_model_amd.bands_probe', bytes(unique_id).hex(), str(rank), str(nranks),
           str(-1 if south is None else south), str(-1 if north is None else north), str(device_index), str(dtype)]
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'TORCHELASTIC_RUN_ID'):
        env.pop(k, None)         # the child is not a rank of this job's process group
    env['PYTHONPATH'] = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + os.pathsep + env.get('PYTHONPATH', '')
    try:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return False, 'probe still running after %g s (killed)' % timeout
    except OSError as e:
        return False, 'probe could not be started: %r' % (e,)
    if r.returncode == 0 and 'BANDS_PROBE_OK' in r.stdout:
        return True, None
    return False, 'probe failed (exit %d): %s' % (r.returncode, (r.stderr or r.stdout)[-400:].strip())


def _negotiate_native(core, layout, rank, device, group):
    """Bring the in-library RCCL transport up on every rank, or on none.  Each step that can fail on one rank alone is
    followed by ONE all_reduce(MIN) of a success flag on the process group that already exists, so that the ranks
    always take the same branch:
      1. librccl resolves and hands out an id (on every rank: the same dlopen the communicator needs);
      2. the group's first rank broadcasts its id;
      3. probe (DYCORE_BANDS_PROBE=0 to skip): a child process per rank runs the whole first contact -- communicator,
         peer connections, two band steps -- between the same devices, bounded by a timeout;
      4. the real communicator + connect, under a watchdog that turns a hang into exit code 86.
    Returns (NativeBands or None, reason it is None)."""
    import os
    import sys
    import torch.distributed as dist
    from .hip_backend import NativeBands, bands_unique_id, DycoreError
    dtype = getattr(core, 'dtype', 'float64')

    first = dist.get_global_rank(group, 0) if group is not None else 0

    def say(msg):
        if rank == first:
            print('climate_model_amd.bands: in-library RCCL transport not used (%s); all ranks use torch.distributed P2P' % msg,
                  file=sys.stderr, flush=True)

    my_id, err = None, None
    try:
        my_id = bands_unique_id(dtype)
    except DycoreError as e:
        err = str(e)
    if not _all_agree(dist, group, device, my_id is not None):
        say(err or 'librccl unavailable on another rank')
        return None, err or 'librccl unavailable on another rank'
    src = first
    ids = [my_id]
    dist.broadcast_object_list(ids, src=src, group=group)
    south, north = layout.south(rank), layout.north(rank)
    timeout = float(os.environ.get('DYCORE_BANDS_TIMEOUT', '240'))
    if os.environ.get('DYCORE_BANDS_PROBE', '1') != '0':
        # the probe needs an id of its own (a communicator consumes one): the first rank draws a second one
        pid = [bands_unique_id(dtype) if rank == src else None]
        dist.broadcast_object_list(pid, src=src, group=group)
        dev_index = device.index if getattr(device, 'index', None) is not None else 0
        ok, why = _probe_native(pid[0], rank, layout.nbands, south, north, dev_index, dtype, timeout)
        if not _all_agree(dist, group, device, ok):
            say(why or 'the probe failed on another rank')
            return None, why or 'the probe failed on another rank'
    nb, why = None, None
    with _Watchdog(timeout, 'ncclCommInitRank / first exchange of the band transport'):
        try:
            nb = NativeBands.rccl(core, ids[0], rank, layout.nbands, south, north)
            nb.connect()
            core.sync()
        except DycoreError as e:
            why = str(e)
            nb = None
    if not _all_agree(dist, group, device, nb is not None):
        if nb is not None:
            nb.close()
        say(why or 'communicator or connect failed on another rank')
        return None, why or 'communicator or connect failed on another rank'
    return nb, None


class DistributedBands:
    """One process = one band.  Drives the Matsuno step in parts with neighbour exchanges
    through torch.distributed (RCCL when the process group's backend is nccl).  With
    overlap=True (default on a GPU) the exchanges run on a second stream behind the rows that
    do not depend on them (_overlapped_step); overlap=False is the plain sequence
    part 0, exchange A, part 1, exchange B, part 2."""

    def __init__(self, core, layout, rank, device, group=None, overlap=None, native=None):
        import torch.distributed as dist
        self.dist = dist
        self.core, self.layout, self.rank = core, layout, rank
        self.group = group
        self.south, self.north = layout.south(rank), layout.north(rank)
        # RCCL process group + a handle on the fused step: the whole band step runs inside the library
        # (hip_backend.NativeBands: ncclSend / ncclRecv issued by the library itself on its comm stream, one call per
        # step sequence; dry, moist and turbulence-coupled handles alike).  The torch.distributed transport below
        # remains for gloo (tests, one-GPU rehearsals), behind DYCORE_BANDS_NATIVE=0, and as the fallback ALL ranks
        # take together when the in-library transport cannot be brought up on any one of them (_negotiate_native).
        self.native, self.native_error = None, None
        if native is None:
            import os
            native = os.environ.get('DYCORE_BANDS_NATIVE', '1') != '0'
        if (native and str(device) != 'cpu' and dist.get_backend(group) == 'nccl' and getattr(core, 'overlap_parts', False)
                and getattr(core, 'flux_from_winds', False) and layout.nbands > 1):
            self.native, self.native_error = _negotiate_native(core, layout, rank, device, group)
        self.halo = BandHalo(core, layout.ny(rank), self.south is not None, self.north is not None)
        self.bufs = _TorchBuffers(self.halo, device)
        on_gpu = str(device) != 'cpu'
        self.staged = dist.get_backend(group) != 'nccl' and on_gpu
        # the split parts exist for the fused step (dry, moist, turbulence-coupled); a handle that steps on the stage
        # kernels keeps the plain sequence
        if overlap is None:
            import os
            overlap = on_gpu and os.environ.get('DYCORE_BANDS_OVERLAP', '1') != '0'
        self.overlap = bool(overlap) and getattr(core, 'overlap_parts', False)
        if self.overlap:
            core._band_halo, core._band_bufs = self.halo, self.bufs
            self.streams = _Streams([core], device)

    def _transport(self, ph):
        """send/recv of the packed buffers with both neighbours, on the current stream."""
        import torch
        dist, b = self.dist, self.bufs
        if self.staged:     # gloo: through host memory
            send, recv = b.send[ph].cpu(), torch.empty(b.recv[ph].shape, dtype=b.recv[ph].dtype)
        else:
            send, recv = b.send[ph], b.recv[ph]
        (ss, sn), (rs, rn) = b.split_send[ph], b.split_recv[ph]
        ops = []
        if self.south is not None:
            ops.append(dist.P2POp(dist.isend, send[:ss], self.south, self.group))
            ops.append(dist.P2POp(dist.irecv, recv[:rs], self.south, self.group))
        if self.north is not None:
            ops.append(dist.P2POp(dist.isend, send[ss:ss + sn], self.north, self.group))
            ops.append(dist.P2POp(dist.irecv, recv[rs:rs + rn], self.north, self.group))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if self.staged:
            b.recv[ph].copy_(recv)

    def connect(self):
        """One transport of the (still meaningless) send buffers into the receive buffers, nothing unpacked:
        RCCL sets a peer connection up at the first send / recv between two ranks, which belongs in front of a
        timed region, not inside it.  Every rank calls it."""
        if self.south is None and self.north is None:
            return
        if self.native is not None:
            self.native.connect()
            self.core.sync()
            return
        import torch
        if self.overlap:
            self.streams.comm.wait_stream(self.streams.main)
            with torch.cuda.stream(self.streams.comm):
                self._transport('A')
            self.streams.main.wait_stream(self.streams.comm)
        else:
            self._transport('A')
        if str(self.bufs.send['A'].device) != 'cpu':
            self.core.sync()
            torch.cuda.synchronize()

    def exchange(self, ph, sub=1):
        """pack, transport, unpack on the current stream."""
        if self.south is None and self.north is None:
            return
        self.core.pack_rows(*_merged(self.halo, ph, sub, 'send'), self.bufs.send[ph].data_ptr())
        self._transport(ph)
        self.core.unpack_rows(*_merged(self.halo, ph, sub, 'recv'), self.bufs.recv[ph].data_ptr())

    def _deliver(self, ph, sub):
        self._transport(ph)
        self.core.unpack_rows(*_merged(self.halo, ph, sub, 'recv'), self.bufs.recv[ph].data_ptr())

    def step_matsuno(self, dt, nsteps=1, with_rho=None):
        core = self.core
        if self.south is None and self.north is None:
            return core.step_matsuno(dt, nsteps, with_rho)
        if self.native is not None:
            return self.native.step_matsuno(dt, nsteps, with_rho)
        if self.overlap:
            return _overlapped_step([core], self.streams, self._deliver, dt, nsteps, with_rho)
        for _ in range(nsteps):
            core.step_begin(with_rho)
            for sub in (0, 1):
                core.substep_part(dt, sub, 0)
                self.exchange('A', sub)
                core.substep_part(dt, sub, 1)
                self.exchange('B', sub)
                core.substep_part(dt, sub, 2)


class LocalBands:
    """All bands inside one process (several handles on the same GPU and stream), exchanging
    through device-to-device copies.  Test vehicle for the band-aware kernels and for the
    overlapped schedule (overlap=True: same stream choreography as DistributedBands)."""

    def __init__(self, cores, layout, device, overlap=True, native=False):
        self.cores, self.layout = cores, layout
        # native: the schedule runs inside the library (dycore_bands_create_local), device copies between the bands
        self.native = None
        if native and len(cores) > 1:
            from .hip_backend import NativeBands
            self.native = NativeBands.local(cores)
            self.overlap = True
            return
        self.halos = [BandHalo(c, layout.ny(r), layout.south(r) is not None, layout.north(r) is not None)
                      for r, c in enumerate(cores)]
        self.bufs = [_TorchBuffers(h, device) for h in self.halos]
        self.overlap = bool(overlap) and len(cores) > 1 and all(c.overlap_parts for c in cores)
        if self.overlap:
            for c, h, b in zip(cores, self.halos, self.bufs):
                c._band_halo, c._band_bufs = h, b
            self.streams = _Streams(cores, device)

    def _deliver(self, ph, sub, host_sync=False):
        """neighbour copies + unpack on the current stream (the packs were queued before it)."""
        n = len(self.cores)
        for r in range(n):
            b = self.bufs[r]
            (rs, rn) = b.split_recv[ph]
            if r > 0:       # from the southern neighbour: its north part
                ss2, sn2 = self.bufs[r - 1].split_send[ph]
                b.recv[ph][:rs].copy_(self.bufs[r - 1].send[ph][ss2:ss2 + sn2])
            if r < n - 1:   # from the northern neighbour: its south part
                ss2, sn2 = self.bufs[r + 1].split_send[ph]
                b.recv[ph][rs:rs + rn].copy_(self.bufs[r + 1].send[ph][:ss2])
        if host_sync:
            import torch
            torch.cuda.synchronize()
        for r in range(n):
            self.cores[r].unpack_rows(*_merged(self.halos[r], ph, sub, 'recv'), self.bufs[r].recv[ph].data_ptr())

    def exchange(self, ph, sub=1):
        n = len(self.cores)
        if n == 1:
            return
        for r in range(n):
            self.cores[r].pack_rows(*_merged(self.halos[r], ph, sub, 'send'), self.bufs[r].send[ph].data_ptr())
        for c in self.cores:
            c.sync()
        self._deliver(ph, sub, host_sync=True)

    def step_matsuno(self, dt, nsteps=1, with_rho=None):
        if self.native is not None:
            return self.native.step_matsuno(dt, nsteps, with_rho)
        if self.overlap:
            return _overlapped_step(self.cores, self.streams, self._deliver, dt, nsteps, with_rho)
        for _ in range(nsteps):
            for c in self.cores:
                c.step_begin(with_rho)
            for sub in (0, 1):
                for c in self.cores:
                    c.substep_part(dt, sub, 0)
                self.exchange('A', sub)
                for c in self.cores:
                    c.substep_part(dt, sub, 1)
                self.exchange('B', sub)
                for c in self.cores:
                    c.substep_part(dt, sub, 2)


def band_grid_arrays(GRF, layout, r):
    """Band r's slice (halo rows included) of the global grid-constant dict."""
    out = {}
    for name, a in GRF.items():
        a = np.asarray(a)
        if a.ndim == 3 and a.shape[0] > 1:      # 2-D horizontal array (n0, n1, 1)
            stgy = 1 if name in ('dxjs', 'dlon_rad') else 0
            out[name] = layout.slice_rows(a, r, stgy)
        else:
            out[name] = a
    return out


def gather_global(layout, band_arrays, name):
    """Reassemble a global [i, j, k] array from per-band local arrays (owned rows only;
    the global halo rows 0 and ny+1(+1) come from the polar bands)."""
    stgy = FIELD_TABLE[name][1]
    parts = []
    for r, a in enumerate(band_arrays):
        lo, hi = layout.owned_rows(r, stgy)
        if r == 0:
            lo = 0
        if r == layout.nbands - 1:
            hi = a.shape[1]
        parts.append(a[:, lo:hi])
    return np.concatenate(parts, axis=1)
