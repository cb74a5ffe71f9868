"""The PGS turn of the two-env constraint kernel is inline asm (csrc/fmj_cons2_rows.inc PGS_QUAD_D / PGS_TURN_S): inside it the compiler pads no hazards, and
the rule that matters there - the DPP operand of a DPP instruction must not be a VGPR a VALU instruction wrote fewer than two wait states earlier - is kept by
the order of the turn's own instructions.  This test compiles the kernel to assembly (hipcc cross-compiles without a GPU) and checks the rule
on what the compiler actually emitted, its own DPP code included (scripts/dpp_hazards.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not on PATH')
def test_dpp_reads_come_two_wait_states_after_valu_writes(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, 'scripts'))
    import dpp_hazards
    out = str(tmp_path/'cons2.s')
    # the flags of _lib.build() for the register row length of the salamander, the fused two-env constraint kernel alone (-DFMJ_DEV_CONS2_FUSED_ONLY)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-fno-slp-vectorize', '-mllvm', '-pragma-unroll-threshold=131072', '-fPIC',
                           '-DFMJ_TU_MAXD=20', '-DFMJ_DEV_CONS2_FUSED_ONLY', '-S', '--cuda-device-only', '-o', out,
                           os.path.join(ROOT, 'farms_mujoco_amd', 'csrc', 'fmj_hip.hip')])
    n, bad = dpp_hazards.check(out)
    text = open(out).read()
    assert 'v_mfma_f32_32x32x2' in text and 'row_newbcast' in text and 'v_permlane16_swap' in text      # the code under test is in the listing
    assert n > 500, n
    assert not bad, bad[:5]
