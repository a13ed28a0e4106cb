#!/usr/bin/env python3
"""What IS a slow placement?    K panels of the n = 100 000 headline problem held at once in one process —
different physical memory each, some fast, some slow (tools/placement_probe.py) — and the SAME product kernel launched REPS + 1
times on each in turn.  Run plain it prints the launch time per panel; run under `rocprofv3 --pmc ... --kernel-trace
--output-format csv` the per-dispatch counter rows of symv_tiles_kernel map back to the panels by dispatch order
(tools/placement_counters_table.py).  The panels stay EMPTY (zeros): only the addresses matter to what is measured.

    python tools/placement_counters.py [K] [REPS]
"""
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiml_amd import device  # noqa: E402
from optiml_amd.datasets import make_blobs  # noqa: E402
from optiml_amd.ml.svm.kernels import gaussian  # noqa: E402
from optiml_amd.opti import KernelQuadratic  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 5
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n, d = 100000, 128
X, y = make_blobs(n, d, seed=0)
ctx = device.get_context()
quads = [KernelQuadratic(X, -np.ones(n), 'svc', gaussian, y=y) for _ in range(K)]
devs = [q.device_problem(ctx) for q in quads]
ms = [dv.time_matvec(reps) for dv in devs]          # reps + 1 launches of symv_tiles_kernel per panel, panel after panel
again = [dv.time_matvec(reps) for dv in devs]
print(json.dumps({'what': 'placement_counters', 'panels': K, 'launches_per_panel_and_sweep': reps + 1, 'sweeps': 2,
                  'ms_per_launch': ms, 'ms_per_launch_second_sweep': again}), flush=True)
