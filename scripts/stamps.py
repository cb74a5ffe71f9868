"""Diagnostic: per-phase cycle shares of the fused step kernel (a -DFMJ_STAMPS build of the library, made on demand:
``python -c "from farms_mujoco_amd import _lib; _lib.build(defines=['-DFMJ_STAMPS'], out='libfmj_hip_stamps.so')"``)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from farms_mujoco_amd import _lib
_lib.SO_PATH = os.path.join(_lib.CSRC, os.environ.get('FMJ_STAMPS_SO', 'libfmj_hip_stamps.so'))
import torch, bench
names = ['emit+drag', 'joints row', 'K', 'C', 'V', 'F+carry', 'S', 'Q', 'M', 'L', 'X', 'Euler', 'facM', 'collide', 'Jrows', 'rowprm', 'Y',
         'A|nwt-start', 'warm|nwt-H', 'PGS|nwt-update', 'qfrc_c', 'nwt-factor', 'nwt-solve', 'nwt-linesearch']
workload = os.environ.get('FMJ_WORKLOAD', 'swim')
for n in (int(a) for a in sys.argv[1:] or ['256', '4096']):
    sim, m, _ = bench.build_sim(n, 1 << 30, 100, 0, 'cuda:0', workload, **({'morphology': os.environ['FMJ_MORPHOLOGY']} if 'FMJ_MORPHOLOGY' in os.environ else {}))
    print('lds bytes per env', sim.physics.kernel_info())
    for _ in range(int(os.environ.get('FMJ_STAMP_WARM', '100')) // 100 + 1):
        sim.step_fused(100)
    torch.cuda.synchronize()
    st = sim.physics.data.qacc[0, :24].cpu().numpy()/100.0
    tot = st.sum()
    print(f'n_envs={n}: cycles/step {tot:.0f}')
    print('  ' + '  '.join(f'{k}:{v:.0f}({100*v/tot:.0f}%)' for k, v in zip(names, st)))
    if workload.startswith('walk'):
        nc = sim.physics.data.ncon.float()
        print(f'  ncon mean {nc.mean().item():.1f} max {nc.max().item():.0f} env0 {nc[0].item():.0f}')
