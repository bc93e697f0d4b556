"""Interleaved A/B of run-time tuning knobs for the finest-level Jacobi sweep (one process, one hierarchy)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy

base = {"strip_slices": 64, "nontemporal": 1, "lds_pad": 32768}
variants = [("base", {}), ("strips 32", {"strip_slices": 32}), ("strips 96", {"strip_slices": 96}),
            ("strips 128", {"strip_slices": 128}), ("strips 256", {"strip_slices": 256}), ("strips 0", {"strip_slices": 0}),
            ("nt 0", {"nontemporal": 0}), ("pad 24576", {"lds_pad": 24576}), ("pad 36864", {"lds_pad": 36864}),
            ("pad 40960", {"lds_pad": 40960})]
with DeviceHierarchy.synthetic(3, 2, 7, c=8, mu1=2, mu2=2) as h:
    res = {name: [] for name, _ in variants}
    for rnd in range(3):
        for name, kw in variants:
            for k, v in {**base, **kw}.items():
                h.set_tuning(k, v)
            res[name].append((h.time_kernel("jacobi", 7, 6), h.time_kernel("residual", 7, 3)))
    for name, _ in variants:
        j = sorted(x[0] for x in res[name]); r = sorted(x[1] for x in res[name])
        print(f"{name:12s} jacobi median {j[1]:.3f} ms (min {j[0]:.3f})   residual median {r[1]:.3f} ms")
