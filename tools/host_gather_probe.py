"""Time the host communicators' allgather_rows / allreduce_sum as the library's exchange callback calls them (no GPU)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench_rendezvous import from_env
comm = from_env(timeout=60.0)
w, r = comm.world_size, comm.rank
chunk = 60416
n = chunk * w
buf = np.zeros(n)
for name, fn in (('allgather_rows', lambda: comm.allgather_rows(buf, r * chunk, (r + 1) * chunk)), ('allreduce_sum', lambda: comm.allreduce_sum(buf[:chunk]))):
    for _ in range(3): fn()
    t = time.time()
    for _ in range(10): fn()
    dt = (time.time() - t) / 10
    if r == 0: print(type(comm).__name__, 'world', w, name, '%.2f ms' % (dt * 1e3), flush=True)
comm.close()
