#!/usr/bin/env python3
"""A second build of libbcqp_hip.so with extra compiler flags for SOME translation units, for before / after measurements with
tools/bench_with_lib.py (the product library is never built this way):

    python tools/build_variant.py NAME bq_gram.hip -DBQ_STREAM_FOLD=1      ->  build/variants/NAME/libbcqp_hip.so

Only the named units are recompiled (into build/variants/NAME/); the others are the product build's objects.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiml_amd import build as B  # noqa: E402


def main():
    name = sys.argv[1]
    units = [a for a in sys.argv[2:] if not a.startswith('-')]
    flags = [a for a in sys.argv[2:] if a.startswith('-')]
    B.build()
    out = os.path.join(ROOT, 'build', 'variants', name)
    os.makedirs(out, exist_ok=True)
    objs = []
    for src in B._sources():
        base = os.path.basename(src)
        if base in units:
            obj = os.path.join(out, base + '.o')
            B._compile(src, obj, extra=flags)
        else:
            obj = os.path.join(B.OBJDIR, base + '.o')
        objs.append(obj)
    target = os.path.join(out, B.LIBNAME)
    subprocess.run(['hipcc', '-shared', '-fPIC', f'--offload-arch={B.ARCH}', '-o', target] + objs + ['-ldl'], check=True)
    print(target)


if __name__ == '__main__':
    main()
