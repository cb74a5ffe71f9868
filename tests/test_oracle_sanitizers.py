"""SURVEY 5 (race / memory-error detection): the CPU oracle's known-answer, contact and solver tests once more under
AddressSanitizer + UndefinedBehaviorSanitizer (`make -C oracle asan`), so that the sanitizer build is exercised by every run of
the CPU suite and not only when somebody remembers the Makefile target.  GPU sanitizers are not available on this pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_under_asan_ubsan():
    if os.environ.get('FMJ_ORACLE_SO'):
        pytest.skip('already inside the sanitizer run')
    asan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip('libasan not installed')
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-s', 'asan'])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1',
               FMJ_ORACLE_SO=os.path.join(ROOT, 'oracle', '_build', 'libfmj_oracle_asan.so'))
    r = subprocess.run([sys.executable, '-m', 'pytest', '-x', '-q', '-m', 'not gpu', '-p', 'no:cacheprovider', 'tests/test_oracle_kat.py', 'tests/test_oracle_contacts.py',
                        'tests/test_golden.py', 'tests/test_cpg_network.py', 'tests/test_oracle_conditioning.py',
                        'tests/test_oracle_solvers.py::test_pgs_newton_cg_agree_on_the_box_bot',
                        'tests/test_oracle_solvers.py::test_pgs_newton_cg_agree_on_random_contact_trees[100]',
                        'tests/test_oracle_solvers.py::test_pgs_newton_cg_agree_on_random_contact_trees[104]'],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert 'runtime error' not in r.stdout + r.stderr and 'AddressSanitizer' not in r.stdout + r.stderr, tail
