"""Interleaved A/B of the resident-blocks-per-CU limit (dynamic LDS padding) for the finest-level Jacobi sweep,
in one process on one hierarchy (cdna_hip_programming.md rule 24)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy

pads = [0, 12288, 16384, 20480, 27306, 32768, 40960, 54613]
with DeviceHierarchy.synthetic(3, 2, 7, c=8, mu1=2, mu2=2) as h:
    res = {p: [] for p in pads}
    for rnd in range(4):
        for p in pads:
            h.set_tuning("lds_pad", p)
            res[p].append(h.time_kernel("jacobi", 7, 8))
    for p in pads:
        v = sorted(res[p])
        print(f"pad {p:6d}  blocks/CU<= {min(8, 163840 // max(p, 1)) if p else 8}  median {v[len(v)//2]:.3f} ms  min {v[0]:.3f}  all {['%.2f' % x for x in res[p]]}")
