"""What rows without a class cost: BASELINE config 4's hierarchy generated with a share of odd rows ("gen_odd_rows": a
reaction term of their own on the diagonal, so the finest level has millions of distinct rows), timed with the escape
dictionary ("row_escape" 1: classes for the frequent rows, the others fetched from the stored matrix by the K-sweep march)
and without it (no row classes: the plain two-sweep pass on 56 B/row).

    python tools/time_escape.py [finest_level=7] [odd rows in 10000 ...=0 1 10 100 300]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd._capi import MgError                       # noqa: E402
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy          # noqa: E402

hi = int(sys.argv[1]) if len(sys.argv) > 1 else 7
shares = [int(a) for a in sys.argv[2:]] or [0, 1, 10, 100, 300]
for odd in shares:
    for escape in (1, 0):
        if odd == 0 and not escape:
            continue
        with DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=50, mu2=50, gen_odd_rows=odd, row_escape=escape) as h:
            info, st = h.level_info(hi), h.level_storage(hi)
            try:
                ms = h.time_kernel("jacobik3", hi, 4)
                pass_text = f"K-sweep pass {ms:7.3f} ms"
            except MgError:
                ms = h.time_kernel("jacobi2!", hi, 4)
                pass_text = f"two-sweep pass {ms:7.3f} ms"
            h.zero_vector(hi, "v")
            h.vcycle(hi, 1)
            h.sync()
            t0 = time.perf_counter()
            res = h.vcycle(hi, 2, residuals=True)
            h.sync()
            dt = (time.perf_counter() - t0) / 2
            print(f"odd rows {odd:5d}/10000, row_escape {escape}: classes {info['row_classes']:3d}, escape rows {st['escape_rows']:10d} "
                  f"of {info['n_global']}; {pass_text}; V(50,50) {1e3 * dt:8.2f} ms = {1 / dt:6.3f} cycles/s; residuals {res[-1]:.3e}", flush=True)
