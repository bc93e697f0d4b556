import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
with DeviceHierarchy.synthetic(3, 2, 7, c=8, mu1=2, mu2=2) as dev:
    for march, shape, seg in ((0, 0, 0), (1, 0, 0), (1, 1, 0), (1, 0, 8), (1, 0, 16), (1, 1, 8), (1, 1, 16)):
        dev.set_tuning("march_sweeps", march); dev.set_tuning("march_shape", shape); dev.set_tuning("fuse_segments", seg)
        print("march", march, "shape", shape, "seg", seg, "jacobi %.3f residual %.3f" % (dev.time_kernel("jacobi", 7, 8), dev.time_kernel("residual", 7, 8)), flush=True)
    dev.set_tuning("fuse_segments", 0)
    dev.set_params(2, 2, 1.0, smoother="rbgs")
    for march in (0, 1):
        dev.set_tuning("march_sweeps", march)
        print("march", march, "rbgs sweep %.3f" % dev.time_kernel("gs", 7, 5), flush=True)
