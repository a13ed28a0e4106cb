"""Host-side AddressSanitizer / UndefinedBehaviorSanitizer run (auxiliary subsystems).  The GPU pool offers no
device-side sanitizer (XNACK off), so what is instrumented is the library's HOST code: `python -m optiml_amd.build --sanitize`
builds lib/asan/libbcqp_hip_asan.so and tests/c/host_checks.c walks the host-only entry points and error paths through it."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CLANG = '/opt/rocm/lib/llvm/bin/clang'   # the consumer must use the same sanitizer runtime as the library (ROCm's clang)


@pytest.mark.skipif(shutil.which('hipcc') is None or not os.path.exists(CLANG), reason='needs hipcc and ROCm clang')
def test_host_entry_points_under_asan_and_ubsan(tmp_path):
    from optiml_amd import build
    lib = build.build(sanitize=True)
    assert lib.endswith('libbcqp_hip_asan.so') and os.path.exists(lib)
    exe = str(tmp_path / 'host_checks')
    src = os.path.join(REPO, 'tests', 'c', 'host_checks.c')
    libdir = os.path.dirname(lib)
    r = subprocess.run([CLANG, '-x', 'c', '-std=c11', '-g', '-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-I',
                        os.path.join(REPO, 'include'), src, '-o', exe, '-L', libdir, '-lbcqp_hip_asan', '-Wl,-rpath,' + libdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0:exitcode=97', UBSAN_OPTIONS='halt_on_error=1:exitcode=98')
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, f'rc={r.returncode}\n{r.stdout}\n{r.stderr[-4000:]}'
    assert 'host_checks ok' in r.stdout
    assert 'ERROR: AddressSanitizer' not in r.stderr and 'runtime error' not in r.stderr, r.stderr[-4000:]
