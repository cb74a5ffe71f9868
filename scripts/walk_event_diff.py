"""Where does an env of the 1000-step walk (BASELINE configs[3]) leave the oracle's walk?  ("Next" item 1.)

HIP and the fp64 oracle run the same trot free, one step per launch.  Per env the FIRST step whose active set (limited joint
sides, contact geoms in order) differs is located, and at that step the oracle is stepped once more from HIP's own fp32 state:

  drift     the oracle, given HIP's state, takes HIP's decision: the two agree on the step, the states had moved apart;
  decision  the oracle, given HIP's state, decides differently from HIP: the step itself differs on identical inputs.

For every differing row the deciding quantity is printed: the row's `pos` (contact distance / limit distance) in the run that has
it - how far past the threshold the run that activates it is - next to how far apart the two states were (max |qpos| difference
and the difference in the row's own `pos` where both runs see it one step earlier).  The same bookkeeping is done for the oracle
with fp32 storage (oracle.fp32_state(), level 2) in place of HIP: the yardstick.

  python scripts/walk_event_diff.py [newton|pgs] [n_envs] [n_steps]   ->  table on stdout, gpurun_out/walk_event_diff_<solver>.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))


def signature_oracle(m, o, e):
    ne, nc = int(o['nefc'][e]), int(o['ncon'][e])
    lim = [(int(m.jnt_dofadr[int(o['efc'][e, r, 5])]), float(o['efc'][e, r, 3])) for r in range(ne) if int(o['efc'][e, r, 4]) == 0]
    con = [int(g) for g in o['contact'][e, :nc, 16]]
    return [d for d, _ in lim], con


def signature_hip(rows, contact, ncon, e):
    kinds = np.ascontiguousarray(rows[e, :, 6]).view(np.int32)
    lim = []
    for k in kinds:
        if k == 0 or (k & 0x40000000):
            break
        lim.append(int(k & 0xffff))
    nc = int(ncon[e])
    gg = np.ascontiguousarray(contact[e, :nc, 15]).view(np.int32)
    return lim, [int(g & 0xffff) for g in gg]


def limit_dists(m, q):
    """distance to each limit side of every limited hinge: {(dof, side): dist}"""
    out = {}
    for j in range(m.njnt):
        if not m.jnt_limited[j] or m.jnt_type[j] == 0:
            continue
        v = q[m.jnt_qposadr[j]]
        out[int(m.jnt_dofadr[j])] = min(v - m.jnt_range[j, 0], m.jnt_range[j, 1] - v)
    return out


def main():
    import torch
    from oracle import oracle
    from farms_mujoco_amd.model import SOLVERS
    from farms_mujoco_amd.physics import BatchedPhysics
    from test_gpu_contacts import _trot_tape, _walker
    solver = sys.argv[1] if len(sys.argv) > 1 else 'pgs'
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    oracle.build()
    m = _walker()
    if solver == 'newton':
        m.solver = SOLVERS['newton']; m.solver_iterations = 100
    tape = _trot_tape(m, n, T).astype(np.float32)
    tape64 = tape.astype(np.float64)
    phys = BatchedPhysics(m, n)
    d = phys.data
    r64 = lambda t: t.cpu().numpy().astype(np.float64)
    d.qpos[:] = torch.as_tensor(np.tile(m.qpos0, (n, 1)), dtype=torch.float32); d.qvel[:] = 0
    q0 = r64(d.qpos)
    runs = {'oracle': dict(q=q0.copy(), v=np.zeros((n, m.nv)), w=np.zeros((n, m.nv))),
            'floor': dict(q=q0.copy(), v=np.zeros((n, m.nv)), w=np.zeros((n, m.nv)))}
    first = {'hip': [None]*n, 'floor': [None]*n}
    drift = {'hip': [], 'floor': []}
    rel = lambda a, b: np.abs(a - b).max(1)/np.abs(b).max(1)
    for t in range(T):
        pre = dict(q=r64(d.qpos), v=r64(d.qvel), w=r64(d.qacc_warmstart))
        d.ctrl[:] = torch.as_tensor(tape[t], device=d.ctrl.device)
        rows, _ = phys.step_debug(want_pgs=False)
        torch.cuda.synchronize()
        rows = rows.cpu().numpy(); con_h = d.contact.cpu().numpy(); ncon_h = d.ncon.cpu().numpy()
        R = runs['oracle']
        pre_o = dict(q=R['q'].copy(), v=R['v'].copy(), w=R['w'].copy())
        o = oracle.step_tf(m, R['q'], R['v'], ctrl=tape64[t], warmstart=R['w'], want_AR=False)
        R.update(q=o['qpos'], v=o['qvel'], w=o['warmstart'])
        F = runs['floor']
        pre_f = dict(q=F['q'].copy(), v=F['v'].copy(), w=F['w'].copy())
        with oracle.fp32_state():
            f = oracle.step_tf(m, F['q'], F['v'], ctrl=tape64[t], warmstart=F['w'], want_AR=False)
        F.update(q=f['qpos'], v=f['qvel'], w=f['warmstart'])
        drift['hip'].append(rel(r64(d.qpos), R['q'])); drift['floor'].append(rel(F['q'], R['q']))
        for e in range(n):
            so = signature_oracle(m, o, e)
            for name, sig, prs in (('hip', signature_hip(rows, con_h, ncon_h, e), pre), ('floor', signature_oracle(m, f, e), pre_f)):
                if first[name][e] is not None or sig == so:
                    continue
                # the oracle (plain fp64) stepped once from the other run's own pre-step state
                p = oracle.step_tf(m, prs['q'][e:e+1], prs['v'][e:e+1], ctrl=tape64[t, e:e+1], warmstart=prs['w'][e:e+1], want_AR=False)
                sp = signature_oracle(m, p, 0)
                kind = 'drift' if sp == sig else 'decision'
                # deciding quantities: rows present in exactly one of (run, oracle free run)
                ev = []
                odist = {int(g): float(dd) for g, dd in zip(o['contact'][e, :int(o['ncon'][e]), 16], o['contact'][e, :int(o['ncon'][e]), 17])}
                pdist = {int(g): float(dd) for g, dd in zip(p['contact'][0, :int(p['ncon'][0]), 16], p['contact'][0, :int(p['ncon'][0]), 17])}
                for g in set(sig[1]) ^ set(so[1]):
                    ev.append(dict(row='contact', geom=g, in_run=g in sig[1], dist_oracle=odist.get(g), dist_oracle_from_run_state=pdist.get(g)))
                lo, lr = limit_dists(m, pre_o['q'][e]), limit_dists(m, prs['q'][e])
                for dof in set(sig[0]) ^ set(so[0]):
                    ev.append(dict(row='limit', dof=dof, in_run=dof in sig[0], dist_oracle=lo.get(dof), dist_run=lr.get(dof)))
                if not ev:
                    ev.append(dict(row='order', run=sig[1], oracle=so[1]))
                first[name][e] = dict(step=t, kind=kind, events=ev, dq=float(np.abs(prs['q'][e] - pre_o['q'][e]).max()),
                                      dv=float(np.abs(prs['v'][e] - pre_o['v'][e]).max()))
    out = dict(solver=solver, n=n, T=T)
    for name in ('hip', 'floor'):
        D = np.array(drift[name])                                  # [T, n]
        fl = first[name]
        steps = np.array([x['step'] if x else T for x in fl])
        pre_flip = [D[max(steps[e] - 1, 0), e] for e in range(n)]   # rel. qpos distance one step before the env's first differing step
        out[name] = dict(first=fl, within_1e4={k: int((D[k - 1] <= 1e-4).sum()) for k in (300, 600, 1000) if k <= T},
                         median_rel={k: float(np.median(D[k - 1])) for k in (100, 300, 600, 1000) if k <= T},
                         no_flip=int((steps == T).sum()), kinds={k: sum(1 for x in fl if x and x['kind'] == k) for k in ('drift', 'decision')},
                         median_rel_before_first_flip=float(np.median(pre_flip)))
        print(f"== {solver} / {name}: envs with no differing active set in {T} steps {out[name]['no_flip']}/{n}; first differences by kind {out[name]['kinds']};"
              f" within 1e-4 {out[name]['within_1e4']}; median rel {out[name]['median_rel']}")
        for e, x in enumerate(fl):
            if x:
                evs = '; '.join(' '.join(f'{k}={v:.3g}' if isinstance(v, float) else f'{k}={v}' for k, v in ev.items()) for ev in x['events'])
                print(f"   env {e:2d} step {x['step']:4d} {x['kind']:8s} |dq| {x['dq']:.2e} |dv| {x['dv']:.2e} rel-before {D[max(x['step'] - 1, 0), e]:.1e} rel-at-end {D[-1, e]:.1e} :: {evs}")
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', f'walk_event_diff_{solver}.json'), 'w'), indent=1, default=float)


if __name__ == '__main__':
    main()
