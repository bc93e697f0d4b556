"""Times the one-sweep kernels of BASELINE config 4's finest level under a list of tuning settings.
    python tools/time_onesweep.py "" "march_sweeps=0" "march_sweeps=0,cls_blocks_per_cu=1000000" ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy          # noqa: E402

hi = int(os.environ.get("MG_FINEST", "7"))
for spec in sys.argv[1:] or [""]:
    tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in spec.split(",") if kv}
    with DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=1, mu2=1, **tune) as h:
        j = h.time_kernel("jacobi", hi, 5)
        r = h.time_kernel("residual", hi, 5)
        print(f"{spec or 'defaults':60s} jacobi {j:.3f} ms  residual {r:.3f} ms", flush=True)
