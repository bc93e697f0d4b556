"""One-sweep kernels on the 1025^3 level with and without row classes (mg_set_tuning "class_sweeps")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
with DeviceHierarchy.synthetic(3, 2, 7, c=8, mu1=2, mu2=2) as dev:
    for cs in (0, 1, 0, 1):
        dev.set_tuning("class_sweeps", cs)
        print(f"class_sweeps {cs}: jacobi {dev.time_kernel('jacobi', 7, 10):.3f} ms  residual {dev.time_kernel('residual', 7, 10):.3f} ms  "
              f"norm2 {dev.time_kernel('norm2', 7, 10):.3f} ms", flush=True)
