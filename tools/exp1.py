import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
import numpy as np, time
with DeviceHierarchy.synthetic_p2(3, 2, 5, c=8, mu1=2, mu2=2, omega=1.0) as dev:
    dev.set_prolongation("p2")
    for restr in ("direct", "table"):
        dev.set_params(2, 2, 1.0, smoother="mcgs", restriction=restr)
        dev.zero_vector(5, "v")
        res = dev.vcycle(5, 8, residuals=True)
        dev.sync(); t0 = time.perf_counter(); dev.vcycle(5, 4); dev.sync(); dt = (time.perf_counter() - t0) / 4
        print(restr, "257^3 lattice V(2,2) mcgs residuals", " ".join("%.3e" % r for r in res), "| %.1f ms per cycle" % (dt * 1e3), flush=True)
