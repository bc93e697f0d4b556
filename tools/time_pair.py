"""Times the fine-level pair pass of BASELINE config 4 under a list of tuning settings (one handle per setting).
    python tools/time_pair.py "fuse_balance=0" "fuse_balance=1" "fuse_balance=1,fuse_balance_snap=0" ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy          # noqa: E402

hi = int(os.environ.get("MG_FINEST", "7"))
for spec in sys.argv[1:] or [""]:
    tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in spec.split(",") if kv}
    with DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=2, mu2=2, **tune) as h:
        ms = h.time_kernel("jacobi2", hi, 5)
        print(f"{spec or 'defaults':50s} jacobi2 {ms:.3f} ms", flush=True)
