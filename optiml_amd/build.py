"""Build libbcqp_hip.so (the HIP/C-ABI library) in-tree with hipcc for gfx950.

    python -m optiml_amd.build [--force]

hipcc cross-compiles without a GPU; the .so is written to optiml_amd/lib/ so that it travels with
the source snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(HERE, 'lib', 'obj')
LIBNAME = 'libbcqp_hip.so'
ARCH = 'gfx950'

CXXFLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-ffp-contract=on', '-Wall',
            '-Wno-unused-function', '-Wno-unused-result', '-Wno-unused-value']


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hs.append(os.path.join(os.path.dirname(HERE), 'include', 'bcqp.h'))
    return hs


def lib_path():
    return os.path.join(LIBDIR, LIBNAME)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, obj, extra=()):
    extra = list(extra) + os.environ.get('BQ_EXTRA_CXXFLAGS', '').split()   # e.g. -DBQ_DIAG_STAMPS for a diagnostic build
    cmd = ['hipcc', '-x', 'hip', '-c', src, '-o', obj] + CXXFLAGS + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    return r.stderr


SAN_FLAGS = ['-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-g', '-O1']


def build(force=False, verbose=False, sanitize=False):
    """sanitize=True: a second library, lib/asan/libbcqp_hip_asan.so, whose HOST code is instrumented with AddressSanitizer +
    UndefinedBehaviorSanitizer (the GPU pool offers no device-side sanitizer: XNACK is off).  It is what tests/test_sanitize.py
    runs the host-only entry points and error paths under; the product library is never built this way."""
    libdir, objdir, libname = LIBDIR, OBJDIR, LIBNAME
    if sanitize:
        libdir = os.path.join(LIBDIR, 'asan')
        objdir = os.path.join(libdir, 'obj')
        libname = 'libbcqp_hip_asan.so'
    os.makedirs(objdir, exist_ok=True)
    hdrs = _headers()
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(objdir, os.path.basename(src) + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for (src, _), warn in zip(jobs, ex.map(lambda a: _compile(*a, extra=SAN_FLAGS if sanitize else ()), jobs)):
                if verbose and warn.strip():
                    print(warn, file=sys.stderr)
    target = os.path.join(libdir, libname)
    if jobs or _stale(target, objs):
        cmd = ['hipcc', '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', target] + objs + ['-ldl']
        if sanitize:
            cmd += ['-fsanitize=address,undefined']
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return target


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True, sanitize='--sanitize' in sys.argv))
