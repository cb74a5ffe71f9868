"""Writes tests/golden/mujoco_<case>.npz: MuJoCo's own outputs (``mujoco.mj_step`` on ``model2mjcf_xml(m)``) for the cases of
tests/mujoco_pin.py, as committed fixtures - the reference cannot travel to the GPU box, its numbers can.
Needs ``mujoco`` (absent in the build container and on the GPU box: run it wherever a wheel exists, commit the .npz files);
tests/test_golden.py then holds the oracle (CPU) and the HIP path (``-m gpu``) to them.

    pip install mujoco && python tests/golden/make_golden_mujoco.py [case ...]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))


def main(names):
    try:
        import mujoco
    except ImportError:
        sys.exit('mujoco is not importable here: `pip install mujoco` on a machine with network access, run this script there and '
                 'commit tests/golden/mujoco_*.npz')
    import mujoco_pin as mp
    from oracle import oracle
    oracle.build()
    for name in names or list(mp.CASES):
        m = mp.case_model(name)
        inp = mp.case_inputs(name, m, oracle)
        one, _ = mp.mujoco_step(m, inp, 1)
        mid, _ = mp.mujoco_step(m, inp, 100)
        long_, _ = mp.mujoco_step(m, inp, mp.N_LONG)
        out = os.path.join(HERE, f'mujoco_{name}.npz')
        np.savez_compressed(out, mujoco_version=str(mujoco.__version__), n_long=mp.N_LONG, exact=int(mp.CASES[name][1]),
                            **{f'in_{k}': v for k, v in inp.items()}, **{f'step1_{k}': v for k, v in one.items()},
                            step100_qpos=mid['qpos'], step100_qvel=mid['qvel'], long_qpos=long_['qpos'], long_qvel=long_['qvel'])
        print('wrote', out, 'ncon', one['ncon'], 'nefc', one['nefc'])


if __name__ == '__main__':
    main(sys.argv[1:])
