#!/usr/bin/env python3
"""bench.py against another build of libbcqp_hip.so (before / after comparisons under rocprofv3; the ABI must match):

    python3 tools/bench_with_lib.py build/old_lib/libbcqp_hip_r02a.so --storage stream --steps 10 --no-cpu --kkt none

Runs bench.py in this process (no exec: under rocprofv3 the GPU is already initialised when the program starts).
"""
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from optiml_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(root, 'bench.py')] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
