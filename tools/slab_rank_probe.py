"""What does ONE slab's GPU do per V-cycle?  A single rank of a `world`-slab decomposition runs alone on the GPU through
the loopback RCCL stand-in (tests/fake_rccl/fake_rccl_loopback.hip: exchanges become device copies of the rank's own
planes, no peers), so its wall time per cycle is the decomposition's compute + launch cost on one GPU of the node -- boundary /
interior split launches, events, the 1/world-size kernels, the replicated coarse levels -- without the link time and
without the artefacts of `world` ranks time-slicing one GPU (tools/slab_overhead_probe.py).  The bound on the speed-up
over one GPU, before any link time, is t_single / t_rank.  Numerical results of the rank are meaningless.

    MG_RCCL_LIBRARY=tests/fake_rccl/libfake_rccl_loopback.so python tools/slab_rank_probe.py [world] [finest] [mu] [k=v,...] [rank]
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd import _capi                              # noqa: E402
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy          # noqa: E402

assert "loopback" in os.environ.get("MG_RCCL_LIBRARY", ""), __doc__
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 7
mu = int(sys.argv[3]) if len(sys.argv) > 3 else 50
tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in (sys.argv[4] if len(sys.argv) > 4 else "").split(",") if kv}
reps = 3 if mu >= 20 else 20
only = int(sys.argv[5]) if len(sys.argv) > 5 else None              # profile runs: this rank only, no single handle


def timed(h):
    h.prepare_cycle(hi)
    h.zero_vector(hi, "v")
    h.vcycle(hi, 2)
    h.sync()
    t0 = time.perf_counter()
    h.vcycle(hi, reps)
    h.sync()
    return (time.perf_counter() - t0) / reps


t_single = float("nan")
if only is None:
    with DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=mu, mu2=mu) as ser:
        t_single = timed(ser)
print(f"single handle: {t_single * 1e3:.2f} ms per V({mu},{mu}) cycle", flush=True)

buf = C.create_string_buffer(128)
_capi.check(_capi.load().mg_comm_unique_id(buf, 128))
worst = 0.0
for rank in (sorted({0, world // 2, world - 1}) if only is None else [only]):
    h = DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=mu, mu2=mu,
                                  comm=lambda hh: hh.set_comm_rccl(rank, world, buf.raw, replicate_below=1 << 22), **tune)
    t = timed(h)
    if os.environ.get("MG_PROBE_LEVELS"):          # where the time goes: one smoother call of mu sweeps per distributed level
        for level in range(hi, 2, -1):
            if h.level_info(level)["replicated"]:
                break
            h.smooth(level, mu)
            h.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                h.smooth(level, mu)
            h.sync()
            print(f"    level {level}: smooth({mu}) {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
            for name in ("jacobik3", "jacobi2", "jacobi"):
                try:
                    print(f"        one launch of {name}: {h.time_kernel(name, level, 5):.3f} ms", flush=True)
                except Exception as exc:                     # noqa: BLE001
                    print(f"        {name}: {exc}", flush=True)
    print(f"rank {rank} of {world} alone: {t * 1e3:.2f} ms per cycle = {world * t / t_single:.3f} x its share of the single handle"
          f" (graph replays {h.counters()['graph_replays']}, tuning {tune})", flush=True)
    worst = max(worst, t)
    h.close()
print(f"slowest rank {worst * 1e3:.2f} ms: overhead {world * worst / t_single:.3f} x, speed-up bound before link time "
      f"{t_single / worst:.2f} x on {world} GPUs")
