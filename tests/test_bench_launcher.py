"""bench.py's N > 1 path without a GPU: the self-spawn command line, the process-group default, the build stamp."""
import argparse
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spawn_ranks_builds_the_torchrun_command(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen['cmd'], seen['env'] = cmd, env
        return 7
    monkeypatch.setattr('subprocess.call', fake_call)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '2', '--steps', '20', '--warmup', '5', '--same-device'])
    rc = bench.spawn_ranks(argparse.Namespace(gpus=2))
    cmd = seen['cmd']
    assert rc == 7                                                   # the parent exits with the children's code
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=2' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'        # the container hostname may not resolve
    assert int(cmd[cmd.index('--master-port') + 1]) > 0
    i = cmd.index(os.path.join(ROOT, 'bench.py'))
    assert cmd[i + 1:] == ['--gpus', '2', '--steps', '20', '--warmup', '5', '--same-device']     # the ranks get the same arguments
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0' and seen['env']['MASTER_ADDR'] == '127.0.0.1'


def test_only_rank_zero_prints_and_gloo_is_the_default():
    """The one JSON line is printed under `if rank == 0:` and the process group defaults to gloo (north_star: no RCCL on this
    path; the group carries a barrier and two scalar reductions)."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert src.count('print(json.dumps(out))') == 1
    head, tail = src.split('print(json.dumps(out))')
    guard = head.rstrip().splitlines()
    assert any(l.strip() == 'if rank == 0:' for l in guard)
    assert "add_argument('--dist-backend', default='gloo'" in src
    assert 'GRAFT_REPO_ROOT' not in src                              # no driver-detecting switch in the measurement script


def test_usable_cores_and_build_id():
    sys.path.insert(0, ROOT)
    import bench
    from farms_mujoco_amd import _lib
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    b = _lib.build_id()
    assert len(b) == 16 and b == _lib.build_id()
