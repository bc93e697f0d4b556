import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
import numpy as np
with DeviceHierarchy.synthetic_p2(3, 2, 5, c=8, mu1=2, mu2=2, omega=1.0) as dev:
    for kind in ("q1", "p2"):
        dev.set_prolongation(kind)
        dev.zero_vector(5, "v")
        res = dev.vcycle(5, 8, residuals=True)
        print(kind, "257^3 lattice V(2,2) mcgs residuals", " ".join("%.3e" % r for r in res), "ratios", " ".join("%.3f" % (res[i+1]/res[i]) for i in range(7)), flush=True)
