"""Dump what the HIP step assembles (rows of H, qfrc_smooth) and what one step produces, next to the fp64 oracle's values
for the same fp32-rounded state, into gpurun_out/hstage_<model>.npz: the input of the error budget of DESIGN section 2
(which part of the single-step qacc error is the entries of H, which the right-hand side, which the fp32 solve).
GPU box: python scripts/dump_hstage.py"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import farms_mujoco_amd.model as mm
    from farms_mujoco_amd.physics import BatchedPhysics
    from farms_mujoco_amd import _lib
    from oracle import oracle
    oracle.build()
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    for maker in ('salamander33', 'eel', 'centipede'):
        m = getattr(mm, maker)()
        n = 16
        rng = np.random.default_rng(5)
        qpos = np.tile(m.key_qpos, (n, 1)); qpos[:, 7:] += rng.uniform(-0.5, 0.5, (n, m.nq - 7))
        quat = rng.normal(size=(n, 4)); qpos[:, 3:7] = quat/np.linalg.norm(quat, axis=1, keepdims=True)
        qvel = rng.normal(size=(n, m.nv))*0.5
        phys = BatchedPhysics(m, n)
        d = phys.data
        d.qpos[:] = torch.as_tensor(qpos, dtype=torch.float32); d.qvel[:] = torch.as_tensor(qvel, dtype=torch.float32)
        q32 = d.qpos.cpu().numpy().astype(np.float64); v32 = d.qvel.cpu().numpy().astype(np.float64)
        rs = ctypes.c_int32()
        H = torch.zeros(n, m.nv, 32, device='cuda'); qf = torch.zeros(n, m.nv, device='cuda')
        c = phys._cdata()
        _lib.check(phys._lib.fmj_forward_debug(phys._ctx, ctypes.byref(c), 0, H.data_ptr(), ctypes.byref(rs), qf.data_ptr(), None))
        torch.cuda.synchronize()
        rs = rs.value
        Hrows = H.cpu().numpy().ravel()[:n*m.nv*rs].reshape(n, m.nv, rs)
        xpos = d.xpos.cpu().numpy(); xquat = d.xquat.cpu().numpy(); xipos = d.xipos.cpu().numpy()
        phys.step(1)
        torch.cuda.synchronize()
        out = dict(Hrows=Hrows, qfrc=qf.cpu().numpy(), q32=q32, v32=v32, qvel1=d.qvel.cpu().numpy(), qpos1=d.qpos.cpu().numpy(),
                   qacc=d.qacc.cpu().numpy(), xpos=xpos, xquat=xquat, xipos=xipos)
        Mo, qso, xpo, xqo, xio = [], [], [], [], []
        for e in range(n):
            o = oracle.forward_debug(m, q32[e], v32[e], ctrl=np.zeros(m.nu))
            Mo.append(o['M']); qso.append(o['qfrc_smooth']); xpo.append(o['xpos']); xqo.append(o['xquat']); xio.append(o['xipos'])
        ref = oracle.step(m, q32, v32, ctrl=np.zeros((n, m.nu)))
        out.update(M_ref=np.array(Mo), qfrc_ref=np.array(qso), qvel1_ref=ref['qvel'], qpos1_ref=ref['qpos'], qacc_ref=ref['qacc'],
                   xpos_ref=np.array(xpo), xquat_ref=np.array(xqo), xipos_ref=np.array(xio))
        np.savez(os.path.join(ROOT, 'gpurun_out', f'hstage_{maker}.npz'), **out)
        print(maker, 'dumped; qvel1 abs err', np.abs(out['qvel1'] - ref['qvel']).max(), flush=True)
        del phys


if __name__ == '__main__':
    main()
