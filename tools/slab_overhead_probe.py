"""How much do the slab mechanics (boundary/interior split launches, events, grouped exchanges, replicated coarse
levels) cost apart from the links?  All `world` ranks run as threads on ONE GPU through the in-process RCCL
stand-in, so the GPU does the same total sweep work as a single handle plus the redundant replicated levels;
the ratio of the two wall times bounds the non-link overhead of the decomposition.

    MG_RCCL_LIBRARY=tests/fake_rccl/libfake_rccl.so python tools/slab_overhead_probe.py [world] [finest] [mu]

With the capturable stand-in and the opt-in graph path (one hipGraphLaunch per cycle and rank instead of ~4800 calls):

    GPU_MAX_HW_QUEUES=24 MG_TEST_TUNE=graph_comm=1 MG_RCCL_LIBRARY=tests/fake_rccl/libfake_rccl_graph.so python tools/...
"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd import _capi                              # noqa: E402
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy          # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 7
mu = int(sys.argv[3]) if len(sys.argv) > 3 else 50
tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("MG_TEST_TUNE", "").split(",") if kv}

with DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=mu, mu2=mu) as ser:
    ser.zero_vector(hi, "v")
    ser.vcycle(hi, 1)
    ser.sync()
    t0 = time.perf_counter()
    ser.vcycle(hi, 2)
    ser.sync()
    t_single = (time.perf_counter() - t0) / 2
print(f"single handle: {t_single * 1e3:.1f} ms per V({mu},{mu}) cycle", flush=True)

buf = C.create_string_buffer(128)
_capi.check(_capi.load().mg_comm_unique_id(buf, 128))
uid = buf.raw
start = threading.Barrier(world)
times, replays, errors = [0.0] * world, [0] * world, []


def rank_main(rank):
    try:
        h = DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=mu, mu2=mu,
                                      comm=lambda hh: hh.set_comm_rccl(rank, world, uid, replicate_below=1 << 22), **tune)
        h.prepare_cycle(hi)
        h.sync()
        start.wait()
        h.zero_vector(hi, "v")
        h.vcycle(hi, 2 if tune.get("graph_comm") else 1)             # (the first graphed cycle is the capture)
        h.sync()
        start.wait()
        t0 = time.perf_counter()
        h.vcycle(hi, 2)
        h.sync()
        times[rank] = (time.perf_counter() - t0) / 2
        replays[rank] = h.counters()["graph_replays"]
        start.wait()
        h.close()
    except Exception as exc:                                     # noqa: BLE001
        import traceback
        traceback.print_exc()
        errors.append(exc)
        start.abort()


if os.environ.get("MG_DUMP_MAPS"):
    # which library an address of a native stack trace belongs to (the profiler's traces print raw addresses): the
    # executable mappings of this process, written before the ranks start
    with open("/proc/self/maps") as src, open(os.environ["MG_DUMP_MAPS"], "w") as dst:
        dst.writelines(line for line in src if " r-xp " in line or "stack" in line)

threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=1100)
assert not errors, errors
t_slabs = max(times)
print(f"{world} slabs on one GPU: {t_slabs * 1e3:.1f} ms per cycle = {t_slabs / t_single:.3f} x the single handle"
      f" (graph replays per rank: {min(replays)}, tuning {tune})")
if "graph" in os.path.basename(os.environ.get("MG_RCCL_LIBRARY", "")):
    fake = C.CDLL(os.environ["MG_RCCL_LIBRARY"])
    fake.fake_rccl_graph_timeouts.restype = C.c_longlong
    print("handshake time-outs:", fake.fake_rccl_graph_timeouts())
