"""Times the one-sweep and the two-sweep Jacobi kernels on one level (default: BASELINE config C4's finest, 1025^3).

    python tools/time_jacobi2.py [finest_level=7] [reps=10]
"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy

hi = int(sys.argv[1]) if len(sys.argv) > 1 else 7
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
out = {}
with DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=2, mu2=2) as dev:
    n = dev.level_info(hi)["n_global"]
    out["rows"] = n
    out["jacobi_ms"] = dev.time_kernel("jacobi", hi, reps)
    print(out, flush=True)
    for shape in ([int(os.environ['J2_SHAPE'])] if 'J2_SHAPE' in os.environ else (0, 1, 2, 3)):
        for seg in ([int(a) for a in sys.argv[3:]] or (0, 1, 3, 4, 6, 8, 11, 13, 16)):
            for nt in (0,):
                dev.set_tuning("fuse_shape", shape)
                dev.set_tuning("fuse_segments", seg)
                dev.set_tuning("fuse_nontemporal", nt)
                ms = dev.time_kernel("jacobi2!", hi, reps)
                out[f"jacobi2_shape{shape}_seg{seg}_nt{nt}_ms"] = ms
                print(f"shape {shape} seg {seg} nt {nt}: {ms:.3f} ms per pair = {ms / 2:.3f} per sweep, "
                      f"{2 * 56 * n / ms / 1e9:.2f} TB/s one-sweep-equivalent", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/time_jacobi2.json", "w"), indent=1)
