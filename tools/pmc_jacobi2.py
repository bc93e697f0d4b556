"""Workload for a rocprofv3 PMC pass over the two-sweep Jacobi kernel only (1025^3, a few launches):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_jacobi2.py [key=value ...]

key=value pairs are mg_set_tuning knobs (fuse_shape, fuse_segments, nontemporal ...).
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy

with DeviceHierarchy.synthetic(3, 2, 7, c=8, mu1=2, mu2=2) as dev:
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        dev.set_tuning(k, int(v))
    print("jacobi2 ms", dev.time_kernel("jacobi2!", 7, 3))
