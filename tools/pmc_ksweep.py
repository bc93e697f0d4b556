"""Workload for a rocprofv3 PMC pass over the K-sweep Jacobi pass only (1025^3, a few launches):

    rocprofv3 --pmc SQ_WAVE_CYCLES ... --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_ksweep.py [key=value ...]

key=value pairs are mg_set_tuning knobs (fuse_k, fuse_k_shape ...); form=N times every step in form N (wrong results).
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy

form = None
with DeviceHierarchy.synthetic(3, 2, int(os.environ.get("MG_FINEST", "7")), c=8, mu1=2, mu2=2) as dev:
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        if k == "form":
            form = v
        else:
            dev.set_tuning(k, int(v))
    hi = dev.finest_level
    print("jacobik3 ms", dev.time_kernel("jacobik3:form" + form if form else "jacobik3!", hi, 3))
