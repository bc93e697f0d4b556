import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
with DeviceHierarchy.synthetic(3, 2, 7, c=8, mu1=2, mu2=2) as dev:
    def t(**kw):
        for k, v in kw.items():
            dev.set_tuning(k, v)
        return dev.time_kernel("jacobi2!", 7, 6)
    for wi in (124, 114, 112):
        for seg in (0, 3, 6, 8, 12, 16, 20, 24, 32, 48, 64):
            print("wi", wi, "seg", seg, "%.3f" % t(fuse_wi=wi, fuse_segments=seg), flush=True)
