"""Times the block pass (mg_jacobiblk.hip.h) on a middle level in its shapes, next to the one-sweep kernel.

    python tools/time_block.py [level=4] [reps=50]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy          # noqa: E402

lvl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
with DeviceHierarchy.synthetic(3, 2, lvl, c=8, mu1=50, mu2=50) as h:
    n = h.level_info(lvl)["n_global"]
    ms = h.time_kernel("jacobi", lvl, reps)
    print(f"level {lvl}, {n} rows: one sweep per launch {1e3 * ms:7.2f} us", flush=True)
    for k in (2, 3, 4):
        for ez in (11, 19):
            h.set_tuning("fuse_block_k", k)
            h.set_tuning("fuse_block_ez", ez)
            ms = h.time_kernel("jacobiblk!", lvl, reps)
            print(f"  block pass, {k} sweeps per launch, {ez} planes per block: {1e3 * ms:7.2f} us = {1e3 * ms / k:6.2f} us per sweep", flush=True)
    h.set_tuning("fuse_block_k", 0)
    h.set_tuning("fuse_block_ez", 0)
    ms = h.time_kernel("jacobiblk!", lvl, reps)
    print(f"  block pass as chosen: {1e3 * ms:7.2f} us", flush=True)
