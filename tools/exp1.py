import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
with DeviceHierarchy.synthetic(3, 2, 7, c=8, mu1=2, mu2=2) as dev:
    dev.set_tuning("fuse_classes", 0)
    for plain, shape, seg in ((1, 1, 0), (2, 1, 0), (2, 2, 0), (2, 4, 0), (2, 4, 8), (2, 4, 16), (2, 1, 8), (2, 2, 8)):
        dev.set_tuning("fuse_plain", plain); dev.set_tuning("fuse_shape", shape); dev.set_tuning("fuse_segments", seg)
        print("plain", plain, "shape", shape, "seg", seg, "%.3f ms per pair" % dev.time_kernel("jacobi2!", 7, 6), flush=True)
