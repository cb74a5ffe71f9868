#!/usr/bin/env python
"""Benchmark of the hot path: env-steps/sec of the fused salamander-swim loop (BASELINE.json configs[1]).

One "step" = one pass of the hot path over the batch: ring-buffer readout (physics2data) -> drag
(SwimmingHandler.step) -> xfrc glue -> controller -> mj_step, for every env of this rank.
Inputs are resident in HBM before the timed region.  One process per GPU; envs are independent, so there
is NO data-path collective: torch.distributed is used only for the barrier and the max-over-ranks time.

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)


def algorithmic_bytes_per_env_step(m, sim=None, workload='swim', sims=None):
    """B_core = 4(nq+nv+nu) read + 4(nq+nv) written; B_log = 4(20 n_links + 4 n_joints + 6 n_s); config 4
    adds 8 nv (warm start read + written) + 4*12 n_contact_sensors.  The mixed workload averages its buckets."""
    def one(m_, sim_):
        n_links = m_.nbody - 1
        n_joints = m_.n_sensor_joints
        ns = len(m_.swimming) if not workload.startswith('walk') else 0
        core = 4*(m_.nq + m_.nv + m_.nu) + 4*(m_.nq + m_.nv)
        log = 4*(20*n_links + 4*n_joints + 6*ns)
        cons = 0
        if workload.startswith('walk'):
            n_cs = len(sim_.task.data.sensors.contacts.names) if sim_ is not None else 0
            cons = 8*m_.nv + 4*12*n_cs
        return core, log, cons
    if workload == 'mixed' and sims:
        parts = [(one(s_.physics.model, s_), s_.physics.n_envs) for s_ in sims]
        tot = sum(n for _, n in parts)
        core, log, cons = (sum(p[i]*n for p, n in parts)/tot for i in range(3))
    else:
        core, log, cons = one(m, sim)
    return dict(core=core, log=log, cons=cons, full=core + log + cons)


def build_sim(n_envs, n_iterations, chunk, env_offset, device, workload='swim', morphology='salamander33', substeps=1):
    """Fused simulation of one morphology.  workload 'swim' = BASELINE configs[1] (water, drag + buoyancy, no contact);
    'walk' = configs[3] (plane contacts + joint limits, PGS, no water)."""
    import torch
    import farms_mujoco_amd.model as mm
    from farms_mujoco_amd.options import SimulationOptions, ArenaOptions, AnimatOptions, WaterOptions
    from farms_mujoco_amd.control import WaveController
    from farms_mujoco_amd.simulation.simulation import Simulation
    from farms_mujoco_amd.data import AnimatData
    if workload.startswith('walk'):       # walk_pairs / walk_hfield / walk_mesh: the same walker with self-collision pairs, on a heightfield, on mesh feet
        m = mm.salamander33(contacts=True, limits=True, spawn_z=0.045, self_collisions=workload in ('walk_pairs', 'walk_pairs_newton'),
                            terrain='hfield' if workload == 'walk_hfield' else 'plane', mesh_feet=workload == 'walk_mesh')
        if workload == 'walk_elliptic_pgs':          # round 5: MuJoCo's PGS with elliptic cones (block update + friction QCQP per contact)
            m.cone = mm.CONES['elliptic']
        if workload == 'walk_noslip':                # round 5: PGS x 50 followed by the noslip post-pass
            m.noslip_iterations = 10
        if workload in ('walk_newton', 'walk_cg', 'walk_elliptic', 'walk_pairs_newton'):     # MuJoCo's Newton / CG solver with its default settings instead of PGS x 50
            m.solver = mm.SOLVERS['cg' if workload == 'walk_cg' else 'newton']; m.solver_iterations = 100
            if workload == 'walk_elliptic':
                m.cone = mm.CONES['elliptic']
    else:
        m = getattr(mm, morphology)() if substeps == 1 else getattr(mm, morphology)(timestep=1e-3/substeps)
    qpos, qvel, psi = mm.synthetic_batch(m, n_envs, seed=0, env_offset=env_offset)
    opts = SimulationOptions(timestep=m.timestep*substeps, n_iterations=n_iterations, num_sub_steps=substeps)
    ctl = WaveController(m, psi, device=device)
    kw = {}
    if workload.startswith('walk'):
        arena = ArenaOptions(water=WaterOptions(height=None, drag=False), ground_height=0.0)
        amp, lag = mm.trot_controller_params(m)
        ctl.amplitude = torch.as_tensor(amp, dtype=torch.float32, device=device)
        ctl.phase_lag = torch.as_tensor(lag, dtype=torch.float32, device=device)
        pairs = [(b, '') for b in m.body_names[1:] if b.endswith('_3') or b.startswith('body_')]
        kw['data'] = AnimatData(m.timestep, chunk, n_envs, m.body_names[1:], m.hinge_joint_names(), contacts=pairs, device=device)
    else:
        arena = ArenaOptions(water=WaterOptions(height=0.0, drag=True, buoyancy=True, viscosity=1.0))
    sim = Simulation.from_sdf(opts, AnimatOptions.from_model(m), arena, model=m, n_envs=n_envs, device=device,
                              controller=ctl, buffer_size=chunk, **kw)
    sim.reset()
    d = sim.physics.data
    d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32)
    d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
    sim.physics.forward(disable_actuation=True)
    return sim, m, (qpos, qvel, psi)


def usable_cores():
    """Host cores this process may actually use: its affinity mask, capped by the cgroup CPU quota when there is one (a
    container that sees every core of the host but is granted a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                                   # cgroup v2: "<quota> <period>" or "max <period>"
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q != 'max':
            n = min(n, max(1, int(int(q)/int(per))))
    except Exception:
        try:                                               # cgroup v1
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()); per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q//per))
        except Exception:
            pass
    return n


def cpu_baseline(m, sim, target_seconds=12.0):
    """The fp64 CPU oracle (C restatement, NOT MuJoCo: the reference's mj_step loop cannot run here, see
    BASELINE.md §2) timed on this box's host cores on a bounded sample of the same workload."""
    import subprocess
    from oracle import oracle
    from farms_mujoco_amd.model import synthetic_batch
    cores = usable_cores()
    n_envs = 64*cores                                                          # >= 64 envs per thread
    so = os.path.join(ROOT, 'oracle', '_build', 'libfmj_oracle_native.so')
    try:
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s', 'native'])
        import ctypes
        oracle._lib = ctypes.CDLL(so)
    except Exception:      # keep the generic build
        oracle.build()
    qpos, qvel, psi = synthetic_batch(m, n_envs, seed=0)
    xp, xq, xi, sd = [], [], [], []
    for e in range(n_envs):
        o = oracle.forward_debug(m, qpos[e], qvel[e])
        s = o['sensordata'].copy(); s[6*(m.nbody - 1) + 3*m.n_sensor_joints:] = 0.0
        xp.append(o['xpos']); xq.append(o['xquat']); xi.append(o['xipos']); sd.append(s)
    st = dict(qpos=qpos, qvel=qvel, xpos=np.array(xp), xquat=np.array(xq), xipos=np.array(xi), sensordata=np.array(sd))
    h = sim.task._callbacks[0].handler
    c = sim.task._controller
    wave = dict(amplitude=c.amplitude.cpu().numpy(), phase_lag=c.phase_lag.cpu().numpy(), env_phase=psi, frequency=c.frequency)
    water = dict(surface=h.water._surface, velocity=h.water._velocity, viscosity=h.water._viscosity, gravity=-9.81,
                 use_buoyancy=h.buoyancy)
    t0 = time.perf_counter()
    oracle.run_fused(m, st, 300, swim=h.swim_dict(), water=water, buffer_size=100, controller=1, wave=wave, n_threads=cores)
    rate = n_envs*300/(time.perf_counter() - t0)                               # calibration pass
    n_steps = int(max(50, min(20000, target_seconds*rate/n_envs)))
    t0 = time.perf_counter()
    oracle.run_fused(m, st, n_steps, swim=h.swim_dict(), water=water, buffer_size=100, controller=1, wave=wave,
                     n_threads=cores)                                          # rows go to a 100-row ring (the ring length does not change what the oracle computes)
    dt = time.perf_counter() - t0
    out = dict(value=n_envs*n_steps/dt, unit='env-steps/s', cores=cores, kind='port',
               sample=f'{n_envs} envs x {n_steps} steps of the same salamander-33 swim workload, fp64 C oracle '
                      f'(not MuJoCo), {cores} pthreads, {dt:.2f} s')
    out['mj_step'] = mujoco_baseline(m, qpos, qvel)
    return out


def mujoco_baseline(m, qpos, qvel, target_seconds=5.0):
    """if ``mujoco`` imports on this box, the reference's own inner call - a bare ``mujoco.mj_step`` loop on the
    exported MJCF of the same model (reference simulation.py:83-89,156), one env, one thread, no drag / readout - is timed as well
    and labelled separately.  It does not import here or on the GPU box of this pipeline: the field then says so."""
    try:
        import mujoco
    except ImportError:
        return {'value': None, 'unit': 'env-steps/s', 'note': 'mujoco is not importable on this box: the reference\'s mj_step loop cannot be timed (BASELINE.md 2)'}
    from farms_mujoco_amd.simulation.mjcf import model2mjcf_xml
    mj = mujoco.MjModel.from_xml_string(model2mjcf_xml(m, fusestatic=False))
    d = mujoco.MjData(mj)
    d.qpos[:] = qpos[0]; d.qvel[:] = qvel[0]
    for _ in range(200):
        mujoco.mj_step(mj, d)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < target_seconds:
        for _ in range(1000):
            mujoco.mj_step(mj, d)
        n += 1000
    dt = time.perf_counter() - t0
    return {'value': n/dt, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'reference',
            'note': f'bare mujoco.mj_step loop (mujoco {mujoco.__version__}), 1 env, 1 thread, {n} steps in {dt:.2f} s; no drag callback, no readout'}


def other_workloads(n_envs, chunk, device):
    """Short measurements of BASELINE configs[3] (walking: limits + contacts + PGS; and with the Newton solver) and configs[4] (eel + centipede buckets) on
    this GPU: env-steps/s over 3000 timed steps (at least one launch) after a warm-up of 1000 (the animals stand and walk by then)."""
    import torch
    from farms_mujoco_amd.simulation.buckets import BucketedSimulation
    res = {}
    for name, warm, steps in (('walk', 1000, 3000), ('walk_newton', 1000, 3000), ('mixed', 1000, 3000)):
        if name == 'mixed':
            sims = [build_sim(n_envs//2, 1 << 30, chunk, 0, device, morphology='eel')[0],
                    build_sim(n_envs - n_envs//2, 1 << 30, chunk, n_envs//2, device, morphology='centipede')[0]]
        else:
            sims = [build_sim(n_envs, 1 << 30, chunk, 0, device, workload=name)[0]]
        batch = BucketedSimulation(sims)
        nl = max(1, steps//chunk)                           # timed launches of `chunk` steps
        for _ in range(max(1, warm//chunk)):
            batch.step_fused(chunk)
        torch.cuda.synchronize()
        evs = []
        t0 = time.perf_counter()
        for _ in range(nl):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            batch.step_fused(chunk)
            e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        for s_ in sims:
            s_.physics.check_invalid_state()
        ms = np.array([a.elapsed_time(b) for a, b in evs])
        res[name] = {'value': n_envs*nl*chunk/dt, 'unit': 'env-steps/s', 'warmup': max(1, warm//chunk)*chunk, 'steps': nl*chunk, 'steps_per_launch': chunk,
                     'launch_ms': {'min': float(ms.min()), 'median': float(np.median(ms)), 'max': float(ms.max())},
                     'config': 'BASELINE configs[3]: salamander-33 walking on a plane' if name == 'walk' else
                               'configs[3] with MuJoCo\'s Newton solver (the reference\'s fallback, mjcf.py:1348-1359) instead of PGS x 50' if name == 'walk_newton' else
                               'BASELINE configs[4]: half eels, half centipedes, one bucket (launch, HIP stream) per morphology, side by side'}
        del sims, batch
    return res


def api_paths(n_envs, chunk, device):
    """The same swimming batch through the API a caller uses (reference simulation.py:148-161): ``Simulation.run()`` - the fused path
    with its status read-back and host sync per chunk, which the headline loop (``step_fused`` back to back) never pays -, the same
    with two sub-steps per iteration (mjcf.py:1187-1192), and the unfused path (``before_step`` / ``after_step`` from the host: one
    launch per operator and physics step - what a run with host callbacks costs).  env-steps/s; an env-step = one mj_step of one env."""
    import torch
    out = {}
    for name, n_it, sub, fused in (('run_fused', 4*chunk, 1, True), ('run_fused_substeps2', 3*chunk, 2, True), ('run_unfused', 300, 1, False)):
        sim, m, _ = build_sim(n_envs, n_it, chunk, 0, device, substeps=sub)
        warm = chunk if fused else 20                          # (iterations; the ring holds `chunk` of them, a fused launch fills it once)
        sim.task.n_iterations = warm; sim.task.sim_iterations = warm*sub
        sim.run(fused=fused)                                   # warm-up through the same call
        sim.task.n_iterations = n_it; sim.task.sim_iterations = n_it*sub
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sim.run(fused=fused)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = (n_it - warm)*sub
        out[name] = {'value': n_envs*steps/dt, 'unit': 'env-steps/s', 'iterations': n_it - warm, 'substeps': sub,
                     'call': f'Simulation.run(fused={fused}), ring of {chunk} rows' + ('' if fused else ': two launches per step - fmj_before_step (rows + drag), then the step with the device controller '
                                                                                                   '(round 4: fmj_physics2data + fmj_drag + torch ctrl write + fmj_step)')}
        del sim
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as FRESH child processes (one per GPU, through
    torch.distributed.run) before this process makes any GPU call, stream their output through, exit with their code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=3000)
    ap.add_argument('--envs-per-gpu', type=int, default=4096)
    ap.add_argument('--chunk', type=int, default=1000,
                    help='steps per fused launch = ring-buffer length.  1000 = the reference logs a whole 1000-iteration run (AnimatData of n_iterations '
                         'rows, task.py:62,158); 100 was the default up to round 4 and is what every same-box A/B of DESIGN.md used.  Longer launches '
                         'amortise prologue / epilogue and - walking - average the uneven contact load of the waves of the one resident round')
    ap.add_argument('--min-seconds', type=float, default=6.0,
                    help='the timed region repeats the --steps block until it lasts at least this long (0: exactly one block); 6 s so that an '
                         'external sampler with a 5 s period sees the GPU busy (VERDICT round 4: with 2 s the driver\'s gpu_busy read 2 %%)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the short walk / mixed measurements appended to the swim line')
    ap.add_argument('--workload', default='swim', choices=['swim', 'walk', 'mixed', 'walk_pairs', 'walk_hfield', 'walk_mesh', 'walk_newton', 'walk_cg', 'walk_elliptic', 'walk_pairs_newton', 'walk_elliptic_pgs', 'walk_noslip'],
                    help='swim = headline (BASELINE configs[1]); walk = configs[3]; mixed = configs[4] eel + centipede; walk_pairs / '
                         'walk_hfield / walk_mesh = the walker with self-collision pairs / on a heightfield / on convex-mesh feet')
    ap.add_argument('--dist-backend', default='gloo',
                    help="process group of the N > 1 run: it carries a barrier and two scalar reductions, no data, so gloo on CPU tensors "
                         "is the default (north_star: no RCCL); 'nccl' remains selectable")
    ap.add_argument('--same-device', action='store_true', help='all ranks use cuda:0 (rehearsal only)')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.same_device:
        local_rank = 0
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        if args.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            # gloo announces its connections on the C stdout ("[Gloo] Rank 0 is connected to ..."): keep stdout for the one JSON line
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group(args.dist_backend)
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with torch.distributed.run'
    device = f'cuda:{local_rank}'
    torch.cuda.set_device(local_rank)

    n_envs = args.envs_per_gpu
    K, W = args.steps, args.warmup
    chunk = max(1, args.chunk)
    n_it = 1 << 30                    # the task only needs iteration < n_iterations; the ring has `chunk` rows
    if args.workload == 'mixed':      # bucketed batching: half the envs are eels, half centipedes, no padding
        sims = [build_sim(n_envs//2, n_it, chunk, rank*n_envs, device, morphology='eel')[0],
                build_sim(n_envs - n_envs//2, n_it, chunk, rank*n_envs + n_envs//2, device, morphology='centipede')[0]]
        sim, m = sims[0], sims[0].physics.model
    else:
        sim, m, _ = build_sim(n_envs, n_it, chunk, env_offset=rank*n_envs, device=device, workload=args.workload)
        sims = [sim]

    from farms_mujoco_amd.simulation.buckets import BucketedSimulation
    batch = BucketedSimulation(sims, overlap=os.environ.get('FMJ_BUCKET_OVERLAP', '1') == '1')

    def run(n):
        """n steps in launches of `chunk` (the last one shorter), a HIP event pair on the launch stream around each."""
        done, evs = 0, []
        while done < n:
            c = min(chunk, n - done)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            batch.step_fused(c)               # mixed: one launch per morphology bucket, side by side on their own streams
            e1.record()
            evs.append((e0, e1, c))
            done += c
        return evs

    # warm-up; its event timings also size the timed region: the K-step block is repeated R times so that the region
    # lasts >= --min-seconds (a single 20-step block is one 0.5 ms launch, which is not a measurement)
    run(max(W, 1))
    evw = run(chunk)                  # one more launch, warm, to size the timed region
    torch.cuda.synchronize()
    est = evw[0][0].elapsed_time(evw[0][1])*1e-3/chunk
    R = max(1, int(np.ceil(args.min_seconds/max(K*est, 1e-9)))) if args.min_seconds > 0 else 1
    if world > 1:                                                   # every rank must time the same number of steps
        t = torch.tensor([R], device=device if args.dist_backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        R = int(t.item())
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = run(K*R)
    torch.cuda.synchronize()
    dt_own = time.perf_counter() - t0               # this rank's own finish time, before it waits for the others
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    from farms_mujoco_amd.sharding import max_over_ranks
    dt = max_over_ranks(dt, device=device if args.dist_backend == 'nccl' else None)
    rank_s = [dt_own]
    if world > 1:                                   # the spread over ranks behind the max (a slow GPU or a late launch shows here)
        t = torch.zeros(world, dtype=torch.float64, device=device if args.dist_backend == 'nccl' else 'cpu')
        t[rank] = dt_own
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        rank_s = [float(x) for x in t.cpu()]
    for s_ in sims:
        s_.physics.check_invalid_state()

    # N > 1: the other sharded BASELINE configurations next to the 4096-per-GPU headline, so that ONE driver run records them
    # (VERDICT round 4 item 6): configs[2] = 8192 swimming salamanders per GPU, configs[4] = the mixed batch, bucket then split (every
    # rank holds its share of the eels and of the centipedes).  Same protocol: barrier, K steps in launches of `chunk`, barrier, MAX.
    b = algorithmic_bytes_per_env_step(m, sim, args.workload, sims)
    info = sim.physics.kernel_info()
    has_constraints = sim.physics.has_constraints
    sharded = None
    if world > 1 and not args.no_extras and args.workload == 'swim':
        sharded = {}
        del batch, sims, sim
        torch.cuda.empty_cache()
        for name, per_gpu in (('configs[2]: swim, 8192 envs per GPU', 8192), ('configs[4]: mixed eel + centipede, bucket then split', n_envs)):
            ck = min(chunk, 250)              # (a ring of 250 rows: 8192 swimmers x 1000 rows would be 34 GB)
            if name.startswith('configs[4]'):
                sm = [build_sim(per_gpu//2, n_it, ck, rank*(per_gpu//2), device, morphology='eel')[0],
                      build_sim(per_gpu - per_gpu//2, n_it, ck, world*(per_gpu//2) + rank*(per_gpu - per_gpu//2), device, morphology='centipede')[0]]
            else:
                sm = [build_sim(per_gpu, n_it, ck, rank*per_gpu, device)[0]]
            bt = BucketedSimulation(sm)
            for _ in range(3):
                bt.step_fused(ck)
            torch.cuda.synchronize()
            tw = time.perf_counter()
            bt.step_fused(ck)
            torch.cuda.synchronize()
            est = max(time.perf_counter() - tw, 1e-6)
            tn = torch.tensor([max(8, int(np.ceil(1.5/est)))], device=device if args.dist_backend == 'nccl' else 'cpu')      # a timed region of >= 1.5 s, the same launches on every rank
            dist.all_reduce(tn, op=dist.ReduceOp.MAX)
            nl = int(tn.item())
            dist.barrier()
            t1 = time.perf_counter()
            for _ in range(nl):
                bt.step_fused(ck)
            torch.cuda.synchronize()
            own = time.perf_counter() - t1
            dist.barrier()
            dts = max_over_ranks(time.perf_counter() - t1, device=device if args.dist_backend == 'nccl' else None)
            for s_ in sm:
                s_.physics.check_invalid_state()
            sharded[name] = {'value': per_gpu*world*nl*ck/dts, 'unit': 'env-steps/s', 'envs_per_gpu': per_gpu, 'n_gpus': world, 'steps': nl*ck,
                             'steps_per_launch': ck, 'timed_region_s': dts, 'this_rank_s': own, 'scaling': 'weak'}
            del bt, sm
            torch.cuda.empty_cache()

    if rank == 0:
        # dominant kernel = the fused step kernel; HIP events on the launch stream around every launch
        full = np.array([e0.elapsed_time(e1)*1e-3 for e0, e1, c in evs if c == chunk] or [e0.elapsed_time(e1)*1e-3*chunk/c for e0, e1, c in evs])
        avg_launch_s = float(full.mean())
        alg_bytes_per_launch = b['full']*n_envs*chunk
        achieved = alg_bytes_per_launch/avg_launch_s/1e9
        # HBM bytes and issue shares from the PMC passes (rocprofv3 cannot run inside this process): the committed summary
        # of the same workload, per env-step, produced as DESIGN.md section 5 describes (scripts/pmc_summary.py)
        traffic, binding, src = None, None, None
        from farms_mujoco_amd import _lib as fmj_lib
        build = fmj_lib.build_id()
        stale = None
        try:
            for tr in json.load(open(os.path.join(ROOT, 'profiles', 'latest_traffic.json'))):
                if tr['workload'] == args.workload and tr['envs'] == n_envs:
                    if tr.get('build_id') != build:        # counters of another build of the kernels are not this run's traffic
                        stale = f"{tr.get('source')} was taken on build {tr.get('build_id')}, this is build {build}"
                        continue
                    traffic = (tr['fetch_bytes_per_env_step'] + tr['write_bytes_per_env_step'])*n_envs*chunk
                    binding = tr.get('binding')
                    src = tr.get('source')
        except Exception:
            pass
        names = {'swim': 'salamander swim (~40 DoF)', 'walk': 'salamander walk on plane (PGS contacts)', 'mixed': 'eel + centipede swim (bucketed)',
                 'walk_pairs': 'salamander walk on plane with self-collision pairs', 'walk_hfield': 'salamander walk on a heightfield',
                 'walk_mesh': 'salamander walk on convex-mesh feet', 'walk_newton': 'salamander walk on plane (Newton solver)', 'walk_cg': 'salamander walk on plane (CG solver)', 'walk_elliptic': 'salamander walk on plane (Newton solver, elliptic cone)',
                 'walk_pairs_newton': 'salamander walk on plane with self-collision pairs (Newton solver)',
                 'walk_elliptic_pgs': 'salamander walk on plane (PGS, elliptic cone)', 'walk_noslip': 'salamander walk on plane (PGS + noslip)'}
        out = {
            'metric': f'env-steps/sec, {names[args.workload]} \u00d7{n_envs} envs, 1/2/4/8 MI355X',
            'value': n_envs*world*K*R/dt, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'repeats': R, 'steps_timed': K*R, 'timed_region_s': dt,
            'rank_seconds': {'min': min(rank_s), 'max': max(rank_s), 'mean': float(np.mean(rank_s)),
                             'note': 'each rank\'s own time from the common start barrier to the end of its last launch; timed_region_s adds the closing barrier'},
            'ms_per_step': dt/(K*R)*1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic', 'build_id': build,
            'config': {'workload': {'swim': f'BASELINE configs[1]: {n_envs}x salamander-33 swimming per GPU (nbody={m.nbody}, '
                                            f'nv={m.nv}, nu={m.nu}), drag+buoyancy, no contact, h=1e-3, travelling-wave position '
                                            f'control, sensor rows logged every step; synthetic model per SURVEY Appendix D except limb '
                                            f'kp 0.1 / inertia 1e-6 (Appendix D values are unstable at h=1e-3: DESIGN 3)',
                                    'walk': f'BASELINE configs[3]: {n_envs}x salamander-33 walking on a plane per GPU, joint limits + '
                                            f'sphere/capsule contacts, pyramidal cone, PGS <= 50 sweeps, link/joint/contact rows logged',
                                    'walk_pairs': f'{n_envs}x salamander-33 walking on a plane, + 16 explicit self-collision pairs (feet / trunk / feet)',
                                    'walk_hfield': f'{n_envs}x salamander-33 walking on a 65 x 65 heightfield (+-5 mm)',
                                    'walk_mesh': f'{n_envs}x salamander-33 walking on a plane on convex-mesh feet (12-vertex hulls)',
                                    'walk_newton': f'{n_envs}x salamander-33 walking on a plane, Newton solver (tolerance 1e-8, <= 100 iterations) instead of PGS',
                                    'walk_cg': f'{n_envs}x salamander-33 walking on a plane, CG solver (tolerance 1e-8, <= 100 iterations) instead of PGS',
                                    'walk_elliptic': f'{n_envs}x salamander-33 walking on a plane, Newton solver with the elliptic friction cone',
                                    'walk_pairs_newton': f'{n_envs}x salamander-33 walking on a plane, 16 explicit self-collision pairs, Newton solver',
                                    'walk_elliptic_pgs': f'{n_envs}x salamander-33 walking on a plane, PGS <= 50 sweeps with elliptic cones (block update + QCQP per contact)',
                                    'walk_noslip': f'{n_envs}x salamander-33 walking on a plane, PGS <= 50 sweeps + 10 noslip iterations',
                                    'mixed': f'BASELINE configs[4]: {n_envs//2}x eel (nv 26) + {n_envs - n_envs//2}x centipede (nv 61) swimming per '
                                             f'GPU, one bucket (launch, HIP stream) per morphology, launched side by side'}[args.workload],
                       'envs_per_gpu': n_envs, 'steps_per_launch': chunk, 'sharding': 'independent envs, no collective',
                       'lds_bytes_per_env': info['lds_bytes_per_env']},
            'launch_ms': {'n': int(full.size), 'min': float(full.min()*1e3), 'median': float(np.median(full)*1e3), 'max': float(full.max()*1e3)},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved/HBM_PEAK_GBS, 'traffic': traffic,
                         'frac_log_only': b['log']*n_envs*chunk/avg_launch_s/1e9/HBM_PEAK_GBS,
                         'frac_traffic': (traffic/avg_launch_s/1e9/HBM_PEAK_GBS) if traffic is not None else None,
                         'traffic_note': ('HBM bytes per launch, FETCH_SIZE + WRITE_SIZE per env-step as counted by the PMC passes '
                                          f'({src}) x envs x steps per launch; algorithmic = {alg_bytes_per_launch}') if traffic is not None else
                                         (f'null: {stale}' if stale else 'null: no PMC summary of this workload / batch size under profiles/'),
                         'kernel': (('fmj_step_cons2_kernel<true, MAXD> (two envs per wave; envs beyond 64 constraint rows finish in fmj_step_kernel<true, MAXD, CONS>)'
                                     if has_constraints else 'fmj_step_dual2_kernel<true, MAXD, WPS> (two envs per wave)') if info['threads_per_env'] == 32
                                    else 'fmj_step_kernel<true, MAXD, CONS> (one env per wave)'),
                         'avg_launch_ms': avg_launch_s*1e3,
                         'algorithmic_bytes_per_env_step': b['full'],
                         'algorithmic_bytes_note': f"B_full = B_core {b['core']} + B_log {b['log']}" + (f" + config-4 terms {b['cons']}" if b['cons'] else '')
                                                   + '; B_core (state + ctrl) stays on chip between the steps of a fused launch, so '
                                                     'frac_log_only counts the row payload alone (SURVEY 8d)',
                         'binding': binding,
                         'frac_note': 'frac prices B_full = B_core + B_log, but B_core (state + ctrl, 860 B for salamander-33) stays on chip between the steps '
                                      'of a fused launch and never crosses HBM; frac_log_only prices the row payload alone, frac_traffic the bytes the PMC '
                                      'counters saw (the rows as stored, dead columns of the 48-byte joint rows included): the two that describe bytes that move',
                         'note': 'the step is bound by dependent-chain latency and VALU issue of the tree recursions, not by HBM '
                                 '(binding: share of SQ_WAVE_CYCLES by SQ counter, profiles/): HBM is the nominal bound (SURVEY 8d)'},
        }
        if sharded is not None:
            out['sharded_configs'] = sharded
        if world == 1 and not args.no_extras and args.workload == 'swim':
            # the other BASELINE configurations, briefly, so that the driver's record carries them too (never the headline)
            try:
                out['other_workloads'] = other_workloads(n_envs, chunk, device)
            except Exception as e:
                out['other_workloads'] = {'error': repr(e)}
        if world == 1 and not args.no_extras and args.workload == 'swim':
            try:
                out['api_paths'] = api_paths(n_envs, chunk, device)
            except Exception as e:
                out['api_paths'] = {'error': repr(e)}
        if world == 1 and not args.no_cpu_baseline and args.workload == 'swim':
            try:
                out['cpu_baseline'] = cpu_baseline(m, sim)
            except Exception as e:      # the baseline is reported, never required
                out['cpu_baseline'] = {'value': None, 'unit': 'env-steps/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': f'failed: {e!r}'}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
