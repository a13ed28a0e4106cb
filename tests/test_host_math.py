"""The kernel maps' exp (optiml_amd/csrc/bq_exp.h) is ONE C99 definition compiled both into the HIP library and, here, into a
host program that compares it with libm's exp: <= 1 ulp over [-746, 0], exact edge values, sane above 0."""
import os
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which('gcc') is None, reason='needs gcc')
def test_kernel_map_exp_is_within_one_ulp_of_libm(tmp_path):
    exe = str(tmp_path / 'exp_check')
    r = subprocess.run(['gcc', '-O2', '-std=c99', '-Wall', '-Werror', '-ffp-contract=off', '-I', os.path.join(REPO, 'optiml_amd', 'csrc'),
                        os.path.join(REPO, 'tools', 'exp_check.c'), '-lm', '-o', exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, '6000000'], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'edges ok' in r.stdout
