"""Round-2 timing of the fine-level kernels at 1025^3 (or another level): the class-coded pair pass with / without
the narrow last column, per launch shape, and the one-sweep class kernels.

    python tools/time_round2.py [finest_level=7] [reps=10]
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy

hi = int(sys.argv[1]) if len(sys.argv) > 1 else 7
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
out = {}
with DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=2, mu2=2) as dev:
    info = dev.level_info(hi)
    n = info["n_global"]
    out["rows"] = n
    out["row_classes"] = info["row_classes"]
    dev.set_tuning("fuse_min_rows", 0)
    for cs in (1, 0):
        dev.set_tuning("class_sweeps", cs)
        out[f"jacobi_cs{cs}_ms"] = dev.time_kernel("jacobi", hi, reps)
        out[f"residual_cs{cs}_ms"] = dev.time_kernel("residual", hi, reps)
    dev.set_tuning("class_sweeps", 1)
    print(out, flush=True)
    for shape in (1, 0, 2, 3):
        for narrow in (1, 0):
            for seg in (0,):
                dev.set_tuning("fuse_shape", shape)
                dev.set_tuning("fuse_narrow", narrow)
                dev.set_tuning("fuse_segments", seg)
                ms = dev.time_kernel("jacobi2!", hi, reps)
                out[f"jacobi2c_shape{shape}_narrow{narrow}_seg{seg}_ms"] = ms
                print(f"shape {shape} narrow {narrow} seg {seg}: {ms:.3f} ms per pair; 25 B/row -> {25 * n / ms / 1e9:.3f} TB/s "
                      f"= {25 * n / ms / 1e9 / 8:.3f} of 8 TB/s", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/time_round2.json", "w"), indent=1)
