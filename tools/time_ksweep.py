"""Times the K-sweep pass (mg_jacobik3d.hip.h) on the finest level of BASELINE config 4 under a list of tuning settings,
next to the two-sweep pass, and whole V(50,50) cycles with and without it.

    python tools/time_ksweep.py [finest_level=7] [reps=6] ["fuse_k=4" "fuse_k=4,fuse_k_shape=1" ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy          # noqa: E402

hi = int(sys.argv[1]) if len(sys.argv) > 1 else 7
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
specs = sys.argv[3:] or ["fuse_k=3", "fuse_k=4", "fuse_k=5", "fuse_k=4,fuse_k_shape=1", "fuse_k=4,fuse_k_shape=2", "fuse_k=4,fuse_k_dpp=0"]
with DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=50, mu2=50) as h:
    n = h.level_info(hi)["n_global"]
    ms = h.time_kernel("jacobi2!", hi, reps)
    print(f"{'two-sweep pass':44s} {ms:8.3f} ms = {ms / 2:6.3f} ms per sweep, {25 * n / ms / 1e6:7.1f} GB/s on 25 B/row", flush=True)
    tuned = {}
    for spec in specs:
        for kv in spec.split(","):
            key, val = kv.split("=")
            if key not in ("form", "skip"):
                h.set_tuning(key, int(val))
                tuned[key] = int(val)
        # sweeps per pass the library takes on a level of this size (mg_capi.hip: sweepsk_max)
        k = min(tuned.get("fuse_k", 5), 3 if n < tuned.get("fuse_k4_min_rows", 0) else 5 if n < tuned.get("fuse_k_small_rows", 1 << 22)
                else 4 if n < tuned.get("fuse_k5_min_rows", 1 << 26) else 5)
        form = next((kv.split("=")[1] for kv in spec.split(",") if kv.startswith("form=")), None)
        skip = next((kv.split("=")[1] for kv in spec.split(",") if kv.startswith("skip=")), None)
        name = "jacobik3" + (":skip" + skip if skip else "") + (":form" + form if form else "!")
        ms = h.time_kernel(name, hi, reps)
        print(f"{spec:44s} {ms:8.3f} ms = {ms / k:6.3f} ms per sweep, {25 * n / ms / 1e6:7.1f} GB/s on 25 B/row", flush=True)
        h.set_tuning("fuse_k_shape", 7)
        h.set_tuning("fuse_k_segments", 0)
        h.set_tuning("fuse_k_dpp", 1)
        h.set_tuning("fuse_k_pf", 1)
    if os.environ.get("MG_SKIP_CYCLES"):
        sys.exit(0)
    for k in (0, 3, 4, 5):
        h.set_tuning("fuse_k", k)
        h.zero_vector(hi, "v")
        h.vcycle(hi, 1)
        h.sync()
        t0 = time.perf_counter()
        res = h.vcycle(hi, 3, residuals=True)
        h.sync()
        dt = (time.perf_counter() - t0) / 3
        print(f"V(50,50) cycles with fuse_k={k}: {1e3 * dt:8.2f} ms per cycle = {1 / dt:6.3f} cycles/s; residuals {res}", flush=True)
