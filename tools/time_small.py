import os, sys
sys.path.insert(0, os.getcwd())
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
for dim, lo, hi in ((2, 1, 3), (2, 1, 2), (3, 0, 1)):
    with DeviceHierarchy.synthetic(dim, lo, hi, c=8, mu1=50, mu2=50) as h:
        n = h.level_info(hi)["n_global"]
        out = []
        for mu in (2, 10, 50, 100, 200):
            h.set_params(mu, mu, 2.0 / 3.0)
            out.append((mu, 1e3 * h.time_kernel("jacobi_small", hi, 50)))
        slope = (out[-1][1] - out[2][1]) / (out[-1][0] - out[2][0])
        print(f"dim {dim}, {n} rows: " + ", ".join(f"{mu} sweeps {us:6.1f} us" for mu, us in out) + f"; {slope:5.2f} us per sweep, {out[2][1] - 50 * slope:5.1f} us fixed", flush=True)
