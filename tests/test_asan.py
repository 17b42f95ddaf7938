"""Host-side AddressSanitizer run of the library's host-only entry points: the arena planners
of the chain executor, descriptor validation and error formatting are plain C++ that runs on the host for every step.
`make asan` (mmnas_amd/csrc/Makefile) builds the sources with --cuda-host-only -fsanitize=address in a few seconds; the
driver of tests/host_plan_driver.py then runs against that build in a child process with the ASan runtime preloaded.
(GPU AddressSanitizer is not available on this pool: device code is checked by the parity suite.)"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'mmnas_amd', 'csrc')


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not on PATH')
def test_host_only_entry_points_under_address_sanitizer():
    b = subprocess.run(['make', '-C', CSRC, 'asan', '-j4'], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-3000:]
    lib = os.path.join(ROOT, 'mmnas_amd', 'lib', 'libmmnas_hip_asan.so')
    rt = subprocess.run(['hipcc', '-print-file-name=libclang_rt.asan-x86_64.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip('ASan runtime of the ROCm clang not found')
    env = dict(os.environ, LD_PRELOAD=rt, MMNAS_LIB_PATH=lib,
               ASAN_OPTIONS='detect_leaks=0:verify_asan_link_order=0:abort_on_error=1:halt_on_error=1')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'host_plan_driver.py')], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and 'HOST_PLAN_OK' in p.stdout, (p.stdout[-1500:], p.stderr[-4000:])
    assert 'AddressSanitizer' not in p.stderr, p.stderr[-4000:]
