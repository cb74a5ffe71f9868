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


def algorithmic_bytes_per_env_step(m):
    """B_full = B_core + B_log (fp32)."""
    n_links = m.nbody - 1
    n_joints = m.n_sensor_joints
    ns = len(m.swimming)
    b_core = 4*(m.nq + m.nv + m.nu) + 4*(m.nq + m.nv)
    b_log = 4*(20*n_links + 4*n_joints + 6*ns)
    return b_core + b_log


def build_sim(n_envs, n_iterations, chunk, env_offset, device, workload='swim', morphology='salamander33'):
    """Fused simulation of one morphology.  workload 'swim' = BASELINE configs[1] (water, drag + buoyancy, no contact);
    'walk' = configs[3] (plane contacts + joint limits, PGS, no water)."""
    import torch
    import farms_mujoco_amd.model as mm
    from farms_mujoco_amd.options import SimulationOptions, ArenaOptions, AnimatOptions, WaterOptions
    from farms_mujoco_amd.control import WaveController
    from farms_mujoco_amd.simulation.simulation import Simulation
    from farms_mujoco_amd.data import AnimatData
    if workload == 'walk':
        m = mm.salamander33(contacts=True, limits=True, spawn_z=0.045)
    else:
        m = getattr(mm, morphology)()
    qpos, qvel, psi = mm.synthetic_batch(m, n_envs, seed=0, env_offset=env_offset)
    opts = SimulationOptions(timestep=m.timestep, n_iterations=n_iterations)
    ctl = WaveController(m, psi, device=device)
    kw = {}
    if workload == 'walk':
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


def cpu_baseline(m, sim, target_seconds=15.0):
    """The fp64 CPU oracle (C restatement, NOT MuJoCo: the reference's mj_step loop cannot run here, see
    BASELINE.md §2) timed on this box's host cores on a bounded sample of the same workload."""
    import subprocess
    from oracle import oracle
    from farms_mujoco_amd.model import synthetic_batch
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = min(cores, 16) if os.environ.get('GRAFT_REPO_ROOT') else cores     # a 1-GPU box grants a 16-core share
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
    oracle.run_fused(m, st, 20, swim=h.swim_dict(), water=water, buffer_size=20, controller=1, wave=wave, n_threads=cores)
    rate = n_envs*20/(time.perf_counter() - t0)                                # calibration pass
    n_steps = int(max(50, min(5000, target_seconds*rate/n_envs)))
    t0 = time.perf_counter()
    oracle.run_fused(m, st, n_steps, swim=h.swim_dict(), water=water, buffer_size=n_steps, controller=1, wave=wave,
                     n_threads=cores)
    dt = time.perf_counter() - t0
    return dict(value=n_envs*n_steps/dt, unit='env-steps/s', cores=cores, kind='port',
                sample=f'{n_envs} envs x {n_steps} steps of the same salamander-33 swim workload, fp64 C oracle '
                       f'(not MuJoCo), {cores} pthreads, {dt:.2f} s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=1000)
    ap.add_argument('--warmup', type=int, default=3000)
    ap.add_argument('--envs-per-gpu', type=int, default=4096)
    ap.add_argument('--chunk', type=int, default=100, help='steps per fused launch (= ring-buffer length)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--workload', default='swim', choices=['swim', 'walk', 'mixed'],
                    help='swim = headline (BASELINE configs[1]); walk = configs[3]; mixed = configs[4] eel + centipede')
    ap.add_argument('--dist-backend', default='nccl', help="'gloo' + --same-device rehearses the N>1 path on a 1-GPU box")
    ap.add_argument('--same-device', action='store_true', help='all ranks use cuda:0 (rehearsal only)')
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
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
            dist.init_process_group(args.dist_backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU with torch.distributed.run'
    device = f'cuda:{local_rank}'
    torch.cuda.set_device(local_rank)

    n_envs = args.envs_per_gpu
    K, W = args.steps, args.warmup
    chunk = max(1, min(args.chunk, K))
    if args.workload == 'mixed':      # bucketed batching: half the envs are eels, half centipedes, no padding
        sims = [build_sim(n_envs//2, K + W + chunk, chunk, rank*n_envs, device, morphology='eel')[0],
                build_sim(n_envs - n_envs//2, K + W + chunk, chunk, rank*n_envs + n_envs//2, device, morphology='centipede')[0]]
        sim, m = sims[0], sims[0].physics.model
    else:
        sim, m, _ = build_sim(n_envs, K + W + chunk, chunk, env_offset=rank*n_envs, device=device, workload=args.workload)
        sims = [sim]

    def run(n):
        done, evs = 0, []
        while done < n:
            c = min(chunk, n - done)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for s_ in sims:
                s_.step_fused(c)
            e1.record()
            evs.append((e0, e1, c))
            done += c
        return evs

    run(W)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    evs = run(K)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    from farms_mujoco_amd.sharding import max_over_ranks
    dt = max_over_ranks(dt, device=device if args.dist_backend == 'nccl' else None)
    for s_ in sims:
        s_.physics.check_invalid_state()

    if rank == 0:
        # dominant kernel = the fused step kernel; HIP events on the launch stream around every launch
        full = [(e0.elapsed_time(e1)*1e-3, c) for e0, e1, c in evs if c == chunk]
        avg_launch_s = float(np.mean([t for t, _ in full])) if full else float('nan')
        b_step = algorithmic_bytes_per_env_step(m)
        alg_bytes_per_launch = b_step*n_envs*chunk
        achieved = alg_bytes_per_launch/avg_launch_s/1e9
        info = sim.physics.kernel_info()
        # HBM bytes per launch from the PMC passes (rocprofv3 cannot run inside this process): the committed
        # summary of the same workload, profiles/latest_traffic.json, produced as DESIGN.md section 5 describes.
        traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, 'profiles', 'latest_traffic.json')))
            if tr['steps_per_launch'] == chunk and tr['envs'] == n_envs and tr.get('workload', 'swim') == args.workload:
                traffic = tr['fetch_bytes'] + tr['write_bytes']
        except Exception:
            pass
        out = {
            'metric': 'env-steps/sec, salamander swim (~40 DoF) \u00d74096 envs, 1/2/4/8 MI355X',
            'value': n_envs*world*K/dt, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': dt/K*1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': {'swim': f'BASELINE configs[1]: {n_envs}x salamander-33 swimming per GPU (nbody={m.nbody}, '
                                            f'nv={m.nv}, nu={m.nu}), drag+buoyancy, no contact, h=1e-3, travelling-wave position '
                                            f'control, sensor rows logged every step',
                                    'walk': f'BASELINE configs[3]: {n_envs}x salamander-33 walking on a plane per GPU, joint limits + '
                                            f'sphere/capsule contacts, pyramidal cone, PGS <= 50 sweeps, link/joint/contact rows logged',
                                    'mixed': f'BASELINE configs[4]: {n_envs//2}x eel (nv 26) + {n_envs - n_envs//2}x centipede (nv 61) swimming per '
                                             f'GPU, one bucket per morphology'}[args.workload],
                       'envs_per_gpu': n_envs, 'steps_per_launch': chunk, 'sharding': 'independent envs, no collective',
                       'lds_bytes_per_env': info['lds_bytes_per_env']},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved/HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_note': 'bytes per launch, FETCH_SIZE + WRITE_SIZE as counted (profiles/), algorithmic = '
                                         f'{alg_bytes_per_launch}',
                         'kernel': ('fmj_step_dual_kernel<true, MAXD> (two envs per wave)' if info['threads_per_env'] == 32
                                    else 'fmj_step_kernel<true, MAXD, CONS> (one env per wave)'),
                         'avg_launch_ms': avg_launch_s*1e3,
                         'algorithmic_bytes_per_env_step': b_step,
                         'note': 'latency/VALU-issue-bound tree recursions (profiles/r01_v11_pmc_summary.txt): HBM is the nominal bound (SURVEY 8d)'},
        }
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
