#!/usr/bin/env python3
"""Benchmark of the Matsuno dycore step on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one full Matsuno step (predictor + corrector) of the dry hydrostatic dycore
over the whole synthetic grid; the metric is BASELINE.json's: cell-updates/s
(cells = nx*ny*nz) at 0.25 deg x 64 levels (1440 x 720 x 64), fp64, inputs resident in HBM.
N > 1 (launched by torch.distributed.run, one rank per GPU) cuts the SAME grid into N
latitude bands with RCCL neighbour send/recv between them: strong scaling.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself:
before anything touches the GPU it runs `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a child process, relays the child's JSON line and exits with its return code.

Rank 0 prints one JSON line (contract in the task statement) with two extra objects:
  roofline     : dominant kernel's compulsory bytes per launch / its mean HIP-event duration
                 in the timed region, against the 8 TB/s HBM3E peak; `kernels` gives the same for
                 every priced kernel of the step, `step_*` for the whole step: the compulsory bytes
                 of the kernels the step actually launches (KERNEL_PASSES x launches) / step time
  cpu_baseline : the CPU restatement (oracle/, "port") timed on this box's host cores on a
                 bounded sample of the same workload (N = 1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of this pool supports dmabuf IPC only: RCCL's peer mappings between the ranks need it
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PASSES_PER_STEP = 130          # 3-D field passes of the reference's stage graph per Matsuno step
# Compulsory 3-D field passes per launch of each kernel = distinct 3-D inputs read once +
# outputs written once (2-D/1-D fields and halos neglected, <2 %).
# Stage kernels: the stage table of SURVEY.md 8(d).  Fused kernels move fewer bytes than the
# stages they replace, so they are priced by what THEY must move, not by the stages' sum;
# corrector launches read the step-start state in addition, hence the .5 averages.
KERNEL_PASSES = {
    'k_continuity': 5, 'k_continuity_reg': 3,   # fused: R UWIND,VWIND W WWIND (UFLX, VFLX are not stored)
    'k_prepare': 15, 'k_uflx_tendency': 13, 'k_vflx_tendency': 13,
    'k_pott_tendency': 6, 'k_euler_forward': 9, 'k_primary_diag': 5, 'k_diag_rho': 6,
    # production (round 3): neither PHI nor POTTVB nor PVTF is stored -- the momentum wavefronts carry the hydrostatic
    # integration, the temperature sweep forms the interface POTT it needs from POTT and PVTFVB of two levels
    'k_momentum_dma': 8.0,       # R UWIND,VWIND,WWIND,POTT,PVTFVB (+2 old in the corrector) W UWIND,VWIND: 7 / 9
    'k_momentum_fused': 10.0,    # its register-staged fallback (nz > 160): the same fields + PHI, PVTF
    # momentum + temperature of a substep in one launch (single GPU): R UWIND,VWIND,WWIND,POTT,PVTFVB (+3 old in the
    # corrector) W UWIND,VWIND,POTT,PVTFVB: 9 / 12 -- the temperature wavefronts read the tiles the momentum wavefronts stage
    'k_substep_fused': 10.5,
    'k_pott_fused': 9.5,         # (DYCORE_LAZY_PHI=0 / experiment builds) R POTT,UWIND,VWIND,WWIND,POTTVB (+1 old) W POTT,PVTFVB,PHI,POTTVB
    'k_temp_canon': 7.75,        # R POTT,UWIND,VWIND,WWIND,PVTFVB (+1 old) W POTT,PVTFVB (+ PHIVB at every 4th interface): 7.25 / 8.25 (turbulence-coupled: + R TURB_T W PHI,PHIVB)
    'k_moist': 8.0,              # R QV,QC,UWIND,VWIND,WWIND (+2 old) W QV,QC
    'k_turb_tendencies': 16.0,   # R UWIND,VWIND,PHI,PHIVB,RHO,RHOVB,KMOM,KHEAT,POTT(,QV,QC) W TURB_U,V,T(,QV,QC)
}


def self_launch(args):
    """--gpus N > 1 without a launcher: start the ranks as a CHILD process (this process has not
    touched the GPU and never will), pass the JSON line through, return the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(('127.0.0.1', 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', '1')
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    if proc.returncode != 0 and env.get('DYCORE_BANDS_NATIVE', '1') != '0' and env.get('DYCORE_BANDS_BACKEND', 'nccl') == 'nccl' \
            and not any(l.startswith('{') for l in proc.stdout.splitlines()):
        # the in-library RCCL transport could not be brought up (its watchdog exits 86; a rank that dies takes the job
        # down): ONE more attempt, in a fresh set of child processes, on the torch.distributed transport
        sys.stderr.write('bench.py: the ranks exited with code %d; retrying once with DYCORE_BANDS_NATIVE=0\n' % proc.returncode)
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            cmd[cmd.index('--master-port') + 1] = str(so.getsockname()[1])
        env['DYCORE_BANDS_NATIVE'] = '0'
        proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    return proc.returncode


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--dlat', type=float, default=0.25, help='grid spacing in degrees (workload)')
    ap.add_argument('--nz', type=int, default=64)
    ap.add_argument('--moist', action='store_true', help='also advect QV/QC (dyn_moist; BASELINE config 5 without radiation); not the headline workload')
    ap.add_argument('--radiation', action='store_true', help='radiation heating in the temperature tendency (i_POTT_radiation) with one radiation call before the timed steps; with --moist this is BASELINE config 5')
    ap.add_argument('--f32', action='store_true', help="working_precision = 'float32' (libdycore_hip_f32.so): a secondary line, never the headline")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=15.0, help='CPU baseline budget')
    return ap.parse_args()


def host_cores():
    """Threads for the CPU baseline: the cgroup CPU quota when one is set, else the affinity
    mask, capped at 16 -- a one-GPU box owns a 16-core share of its host although the mask
    shows every core of the machine.  DYCORE_CPU_THREADS overrides."""
    env = os.environ.get('DYCORE_CPU_THREADS')
    if env:
        return int(env)
    n = len(os.sched_getaffinity(0))
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return min(n, 16)


def cpu_baseline(GR, state, budget_s):
    """Times the CPU restatement on the same workload for about `budget_s` seconds."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from oracle_binding import Oracle, ORACLE_FIELDS
    ncores = host_cores()
    need_gb = 46 * (GR.nx + 3) * (GR.ny + 3) * (GR.nz + 1) * 8 / 1e9
    avail_gb = 0.0
    for line in open('/proc/meminfo'):
        if line.startswith('MemAvailable'):
            avail_gb = int(line.split()[1]) / 1e6
    if avail_gb < 1.5 * need_gb + 8:
        return {'value': None, 'unit': 'cell-updates/s', 'cores': ncores, 'kind': 'port',
                'sample': 'skipped: %.0f GB host memory needed, %.0f GB available' % (need_gb, avail_gb)}
    o = Oracle(GR.nx, GR.ny, GR.nz, GR.GRF['CPU'], {n: state[n] for n in state if n in ORACLE_FIELDS},
               pair_top=GR.nl.pair_top, i_turb=0, nthreads=ncores, fill_nan=False)
    for n in ('WWIND', 'POTTVB'):
        if n not in state:
            o.F[n][...] = 0.
    o.primary_diag()
    t0 = time.perf_counter()
    o.step_matsuno(GR.dt, 1, with_rho=False)          # first touch of every array + warm-up
    t_first = time.perf_counter() - t0
    nsteps = int(max(1, min(20, budget_s / max(t_first, 1e-3))))
    t0 = time.perf_counter()
    o.step_matsuno(GR.dt, nsteps, with_rho=False)
    t = (time.perf_counter() - t0) / nsteps
    cells = GR.nx * GR.ny * GR.nz
    return {'value': cells / t, 'unit': 'cell-updates/s', 'cores': ncores, 'kind': 'port',
            'sample': '%d Matsuno steps of the full %dx%dx%d grid (C/OpenMP restatement of the '
                      'reference CPU kernels, %.2f s/step, after 1 untimed step)' % (nsteps, GR.nx, GR.ny, GR.nz, t)}


def main():
    args = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d'
                         % (args.gpus, world, args.gpus))
    import torch
    import torch.distributed as dist
    from climate_model_amd import Namelist, Grid, ModelFields, HipDycore
    from climate_model_amd.bands import BandLayout, DistributedBands, band_grid_arrays
    from climate_model_amd.fields import FIELD_TABLE
    from climate_model_amd.synthetic import initialize_fields, initialize_moisture

    # rehearsal switches (tests only): all ranks on one GPU, gloo instead of RCCL
    backend = os.environ.get('DYCORE_BANDS_BACKEND', 'nccl')
    if os.environ.get('DYCORE_BANDS_ONE_GPU'):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)

    # ---- workload: synthetic dry state on the global grid, this rank's latitude band ----
    GR = Grid(Namelist(nz=args.nz, dlat_deg=args.dlat, dlon_deg=args.dlat))
    L = BandLayout(GR.ny, world)
    F = ModelFields(GR)
    initialize_fields(GR, F)        # index-hash noise: identical on every rank
    names = ['HSURF', 'COLP', 'UWIND', 'VWIND', 'POTT', 'WWIND', 'POTTVB']
    if args.moist:
        initialize_moisture(GR, F)
        names += ['QV', 'QC', 'SLHFLX']
    if args.radiation:
        F.host['SOILTEMP'][:, :, 0] = 295 - np.sin(GR.lat_rad[:, :, 0]) ** 2 * 30         # srfc_main.py:57
        F.host['SURFALBEDLW'][...] = 0.
        F.host['SURFALBEDSW'][...] = 0.2
        if not args.moist:
            F.host['QC'][...] = 0.
            names += ['QC']
        names += ['SOILTEMP', 'SURFALBEDLW', 'SURFALBEDSW']
    state = {n: F.host[n] for n in names}
    core = HipDycore(GR.nx, L.ny(rank), GR.nz, pair_top=GR.nl.pair_top, device=local_rank,
                     ny_global=GR.ny, j_offset=L.j_offset(rank), stream=stream.cuda_stream, i_moist=int(args.moist),
                     i_radiation=int(args.radiation), dtype=np.float32 if args.f32 else np.float64)
    esize = 4 if args.f32 else 8
    core.set_grid(band_grid_arrays(GR.GRF['HIP'], L, rank))
    for n in names:
        core.upload(n, L.slice_rows(state[n], rank, FIELD_TABLE[n][1]))
    core.primary_diag()
    if args.radiation:
        from climate_model_amd.radiation import Radiation, band_radiation
        core.secondary_diag()
        band_radiation(Radiation(GR), GR, L, rank, core)
    core.sync()
    bands = DistributedBands(core, L, rank, torch.device('cuda', local_rank)) if world > 1 else None
    if bands is not None:
        bands.connect()             # peer connections up before anything is timed, whatever --warmup says

    def run(nsteps):
        if bands is None:
            core.step_matsuno(GR.dt, nsteps)
        else:
            bands.step_matsuno(GR.dt, nsteps)

    def fence():
        torch.cuda.synchronize()
        core.sync()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    run(args.warmup)
    # timed region: HIP events around the dominant kernel only (level 2) -- two events per launch of
    # EVERY kernel cost 1 % of the step on the full grid and 20 % on one of 8 latitude bands
    core.profile(2)
    fence()
    t0 = time.perf_counter()
    run(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    stats = core.kernel_stats()
    # per-kernel breakdown: a few more steps with every launch bracketed, outside the timed region
    core.profile(1)
    nbreak = max(1, min(args.steps, 5))
    run(nbreak)
    fence()
    bstats = core.kernel_stats()
    # a moist handle's whole-substep launch advances QV / QC as well (round 3; no k_moist launch then): + R QV,QC (+2 old in
    # the corrector) W QV,QC
    passes = dict(KERNEL_PASSES)
    if args.moist and not bstats.get('k_moist', (0, 0))[0]:
        passes['k_substep_fused'] = KERNEL_PASSES['k_substep_fused'] + 5.0
    breakdown = {k: tot / nbreak for k, (n, tot) in bstats.items() if n}
    launches_per_step = {k: n / nbreak for k, (n, tot) in bstats.items() if n}
    core.profile(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # measured device-to-device copy ceiling of this GPU (reported next to the 8 TB/s spec peak)
    copy_gbs = None
    try:
        nbytes = min(core.field_shape('UWIND')[0] * core.field_shape('UWIND')[1] * core.field_shape('UWIND')[2] * esize, 1 << 30)
        a = torch.empty(nbytes // 8, dtype=torch.float64, device='cuda')
        b = torch.empty_like(a)
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            b.copy_(a)
        e1.record()
        e1.synchronize()
        copy_gbs = 2 * nbytes * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
    except Exception:
        copy_gbs = None

    # sanity: the state must still be finite (a diverged run is not a measurement)
    u = core.download('UWIND')
    if not np.isfinite(u[1:GR.nx + 1, 1:-1]).all():
        raise SystemExit('non-finite UWIND after the timed steps')

    cells = GR.nx * GR.ny * GR.nz
    ms_per_step = 1e3 * elapsed / args.steps
    value = cells * args.steps / elapsed
    band_cells = GR.nx * L.ny(rank) * GR.nz
    # dominant kernel on this rank
    dom, dom_n, dom_ms = None, 0, 0.0
    for k, (n, tot) in stats.items():
        if n and tot > dom_ms and k in passes:
            dom, dom_n, dom_ms = k, n, tot
    roof = None
    if dom:
        per_launch_bytes = passes[dom] * band_cells * esize
        # a band runs the kernel as two or three launches per substep (inner rows behind the exchanges, edge
        # rows between them): price them as the one pass over the band they are
        if world > 1:
            dom_n = 2 * args.steps
        avg_s = dom_ms / dom_n / 1e3
        ach = per_launch_bytes / avg_s / 1e9
        # whole step: the compulsory bytes of the kernels it actually launched (this rank's band), from the
        # per-launch breakdown pass above
        kern, step_bytes = {}, 0.0
        for k, ms in breakdown.items():
            if k not in passes:
                continue
            nl_ = launches_per_step[k]
            gb = passes[k] * band_cells * esize / 1e9
            if world > 1 and k in ('k_momentum_dma', 'k_substep_fused'):
                nl_ = 2.0               # the pieces of a substep (inner tile rows in one or two launches, edge rows) = one pass over the band
            step_bytes += gb * nl_
            kern[k] = {'ms_per_step': ms, 'launches_per_step': launches_per_step[k], 'compulsory_GB_per_pass': gb,
                       'achieved_GBs': gb * nl_ / (ms * 1e-3), 'frac': gb * nl_ / (ms * 1e-3) / HBM_PEAK_GBS}
        step_ach = step_bytes / (ms_per_step * 1e-3)
        traffic, traffic_src = None, None
        tfile = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
        if os.path.exists(tfile) and world == 1:
            try:
                tj = json.load(open(tfile))
                if tj.get('kernel') == dom and tj.get('grid') == [GR.nx, GR.ny, GR.nz]:
                    traffic = tj.get('bytes_per_launch')
                    traffic_src = 'stored PMC profile, not measured in this run: %s' % tj.get('source', 'profiles/traffic_latest.json')
            except Exception:
                traffic = None
        roof = {'bound': 'hbm', 'kernel': dom, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': ach / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                'algorithmic_bytes_per_launch': per_launch_bytes, 'avg_launch_ms': dom_ms / dom_n,
                'launches': dom_n,
                'step_compulsory_GB': step_bytes, 'step_achieved': step_ach, 'step_frac': step_ach / HBM_PEAK_GBS,
                'step_survey_130pass_GBs': PASSES_PER_STEP * esize * value / 1e9 / world,
                'measured_copy_GBs': copy_gbs,
                'kernels': kern,
                'kernels_ms_per_step': breakdown}

    line = {
        'metric': 'cell-updates/sec per Matsuno step at 0.25deg x 64L (achieved HBM GB/s in roofline)',
        'value': value, 'unit': 'cell-updates/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': ms_per_step, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32' if args.f32 else 'f64', 'data': 'synthetic',
        'config': {'workload': ('moist (QV, QC advected) ' if args.moist else 'dry ') + ('with radiative heating ' if args.radiation else '') + 'hydrostatic dycore, Matsuno step, %g deg lat-lon (%d x %d) x %d sigma levels, '
                               'dt=%.4g s' % (args.dlat, GR.nx, GR.ny, GR.nz, GR.dt),
                   'cells': cells, 'parallelism': 'latitude bands x%d' % world if world > 1 else 'single GPU',
                   'exchange': (('overlapped, 2 per substep' if bands.overlap else 'plain, 2 per substep') +
                                (', RCCL send/recv issued by the library' if getattr(bands, 'native', None) is not None else
                                 ', torch.distributed P2P (%s)' % backend) +
                                ((' -- in-library transport refused: %s' % bands.native_error) if getattr(bands, 'native_error', None) else ''))
                               if bands is not None else None},
        'roofline': roof,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.moist and not args.radiation and not args.f32:
        core.close()
        try:
            line['cpu_baseline'] = cpu_baseline(GR, state, args.cpu_seconds)
        except Exception as e:        # the baseline is a report, never a reason to lose the GPU number
            line['cpu_baseline'] = {'value': None, 'unit': 'cell-updates/s', 'cores': None, 'kind': 'port',
                                    'sample': 'failed: %r' % (e,)}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
