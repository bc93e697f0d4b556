"""Full-size slab check on ONE GPU: the 1025^3, 6-level hierarchy split over `world` ranks (threads of this
process, in-process RCCL stand-in from tests/fake_rccl), two V(2,2) cycles, residual norms compared with a
single-handle run.  Not a timing: all ranks share one GPU.  Exercises the slab index arithmetic (planes of
1,050,625 rows, 135 M rows per rank, lead regions of the symmetric storage, replicated coarse levels) at the
headline size.

    MG_RCCL_LIBRARY=tests/fake_rccl/libfake_rccl.so python tools/full_scale_slab_check.py [world] [finest_level]
"""
import ctypes as C
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multigrid_dolfinx_amd import _capi                              # noqa: E402
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy          # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 7
assert "fake_rccl" in os.environ.get("MG_RCCL_LIBRARY", ""), "set MG_RCCL_LIBRARY to the stand-in"

with DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=2, mu2=2) as ser:
    ser.zero_vector(hi, "v")
    want = ser.vcycle(hi, 2, residuals=True)
    want_f = ser.norm2(hi, "f")
print("single handle:", want, want_f, flush=True)

buf = C.create_string_buffer(128)
_capi.check(_capi.load().mg_comm_unique_id(buf, 128))
uid = buf.raw
got, errors = [None] * world, []


def rank_main(rank):
    try:
        h = DeviceHierarchy.synthetic(3, 2, hi, c=8, mu1=2, mu2=2,
                                      comm=lambda hh: hh.set_comm_rccl(rank, world, uid, replicate_below=1 << 22))
        info = h.level_info(hi)
        h.zero_vector(hi, "v")
        res = h.vcycle(hi, 2, residuals=True)
        got[rank] = (res, h.norm2(hi, "f"), info["n_local"], info["row0"])
        h.close()
    except Exception as exc:                                     # noqa: BLE001
        import traceback
        traceback.print_exc()
        errors.append(exc)


threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
for t in threads:
    t.start()
for t in threads:
    t.join(timeout=900)
assert not any(t.is_alive() for t in threads), "a rank is stuck"
assert not errors, errors
rows = sum(g[2] for g in got)
assert rows == (8 * 2 ** hi + 1) ** 3, rows
for rank, (res, fn, nloc, row0) in enumerate(got):
    assert np.all(np.abs(res - want) <= 1e-12 * want), (rank, res, want)
    assert abs(fn - want_f) <= 1e-12 * want_f
print(f"{world} slabs ({[g[2] for g in got]} rows): residuals {got[0][0]} match the single-handle run: OK")
