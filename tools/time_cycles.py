"""Whole V(mu,mu) cycles of BASELINE config 4 (or another finest level) under a list of tuning settings.

    python tools/time_cycles.py [finest_level=7] [mu=50] ["fuse_k=4,fuse_k_shape=1" ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy          # noqa: E402

hi = int(sys.argv[1]) if len(sys.argv) > 1 else 7
mu = int(sys.argv[2]) if len(sys.argv) > 2 else 50
specs = sys.argv[3:] or ["fuse_k=0", "fuse_k=3", "fuse_k=4"]
with DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=mu, mu2=mu) as h:
    for spec in specs:
        for kv in spec.split(","):
            key, val = kv.split("=")
            h.set_tuning(key, int(val))
        h.zero_vector(hi, "v")
        h.vcycle(hi, 1)
        h.sync()
        t0 = time.perf_counter()
        res = h.vcycle(hi, 3, residuals=True)
        h.sync()
        dt = (time.perf_counter() - t0) / 3
        print(f"V({mu},{mu}) {spec:40s}: {1e3 * dt:8.2f} ms per cycle = {1 / dt:6.3f} cycles/s; residuals {res}", flush=True)
