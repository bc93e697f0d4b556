"""One-off check at a size that is not a multiple of the tile: c = 6 -> 769^3 unknowns on the finest of 6 levels.
Two V(3,3) cycles (a pair + a single sweep per leg) with the paired class-coded smoother, with the paired plain
smoother and with single sweeps must give bit-identical iterates: compared through the residual norms and ||v||,
whose reductions are deterministic for identical vectors.

    python tools/fused_vs_single_at_scale.py [c=6] [finest=7]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy

c = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 7
out = []
for name, tune in (("single sweeps", {"fuse_sweeps": 0}), ("pairs, rows", {"fuse_classes": 0}), ("pairs, classes", {})):
    with DeviceHierarchy.synthetic(3, 2, hi, c=c, mu1=3, mu2=3, **tune) as dev:
        info = dev.level_info(hi)
        dev.zero_vector(hi, "v")
        res = dev.vcycle(hi, 2, residuals=True)
        out.append((name, list(res), dev.norm2(hi, "v")))
        print(name, info["n_global"], info["row_classes"], res, out[-1][2], flush=True)
assert all(o[1] == out[0][1] and o[2] == out[0][2] for o in out[1:]), "iterates differ"
print("identical: OK")
