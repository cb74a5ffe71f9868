"""Diagnostic (-DFMJ_STAMPS build): spread of the wave lifetimes of one launch of the two-env constraint kernel - a launch lasts as long
as its slowest wave, and with every wave resident from the start nothing rebalances.  usage: python scripts/wave_spread.py [workload]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from farms_mujoco_amd import _lib
_lib.SO_PATH = os.path.join(_lib.CSRC, 'libfmj_hip_stamps.so')
import numpy as np, torch, bench
wl = sys.argv[1] if len(sys.argv) > 1 else 'walk'
sim, m, _ = bench.build_sim(4096, 1 << 30, 100, 0, 'cuda:0', wl)
for k in range(14):
    sim.step_fused(100)
    torch.cuda.synchronize()
    t = sim.physics.data.qacc[:, m.nv - 1].cpu().numpy()
    nc = sim.physics.data.ncon.cpu().numpy()
    if k >= 12:
        solo = sim.physics.data.qacc[:, m.nv - 2].cpu().numpy(); rows = sim.physics.data.qacc[:, m.nv - 3].cpu().numpy()
        print('  corr(lifetime, SOLO steps) %.2f  corr(lifetime, sum of rows) %.2f;  SOLO steps per wave: mean %.1f max %.0f;  lifetime of waves with 0 SOLO steps %.0f, with > 50: %.0f' % (
              np.corrcoef(t, solo)[0, 1], np.corrcoef(t, rows)[0, 1], solo.mean(), solo.max(), t[solo == 0].mean() if (solo == 0).any() else 0, t[solo > 50].mean() if (solo > 50).any() else 0))
        A_ = np.stack([np.ones_like(t), solo, rows], 1); coef = np.linalg.lstsq(A_, t, rcond=None)[0]; print('  lifetime ~ %.0f + %.0f * SOLO steps + %.0f * rows' % tuple(coef), ' residual std %.0f' % np.std(t - A_ @ coef))
    if k >= 10:
        heavy = nc > 8
        print(f'launch {k}: wave lifetime cycles mean {t.mean():.0f} p10 {np.percentile(t,10):.0f} p50 {np.median(t):.0f} p90 {np.percentile(t,90):.0f} max {t.max():.0f}  '
              f'mean/max {t.mean()/t.max():.2f};  envs with > 8 contacts at the end: {heavy.mean()*100:.0f}% (their waves: mean {t[heavy].mean():.0f}, the others {t[~heavy].mean():.0f})')
