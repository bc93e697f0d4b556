"""Runs in its own process with MG_RCCL_LIBRARY pointing at tests/fake_rccl/libfake_rccl.so: `world` threads
play the ranks of the slab decomposition on one GPU, going through the library's real RCCL code path
(mg_set_comm, grouped ncclSend/ncclRecv on the communication stream, all-reduce, grouped broadcasts) with
asynchronous, event-ordered copies -- unlike the host-staged callback transport, which synchronises every
exchange.  Results must equal a single-handle run bit for bit."""
import ctypes as C
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from multigrid_dolfinx_amd import _capi                              # noqa: E402
from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy          # noqa: E402


def main(world, dim, lo, hi, c, mu, replicate_below, overlap):
    assert "fake_rccl" in os.environ.get("MG_RCCL_LIBRARY", "")
    buf = C.create_string_buffer(128)
    _capi.check(_capi.load().mg_comm_unique_id(buf, 128))
    uid = buf.raw
    results, errors = [None] * world, []
    tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("MG_TEST_TUNE", "").split(",") if kv}
    bar = threading.Barrier(world)

    def phase(h):
        # The capturable stand-in's receives spin ON THE DEVICE until the peer's send is enqueued, and calls that wait for
        # the whole device (hipFree in level set-up, graph destruction) wait for those spins too -- threads of one process
        # share the device, ranks of a real job do not.  So the ranks line up between phases: nobody frees memory while a
        # peer already waits for a message this rank has not enqueued yet.
        h.sync()
        bar.wait(timeout=200)

    def rank_main(rank):
        try:
            def comm(h):
                h.set_tuning("overlap", overlap)
                h.set_tuning("overlap_min_rows", 0)     # tiny grids: still take the overlapped sweep path
                h.set_comm_rccl(rank, world, uid, replicate_below=replicate_below)
            h = DeviceHierarchy.synthetic(dim, lo, hi, c=c, mu1=mu, mu2=mu, comm=comm, **tune)
            info = h.level_info(hi)
            assert not info["replicated"] and info["n_local"] < info["n_global"]
            if os.environ.get("MG_TEST_EXPECT_KSLAB"):
                # the K-sweep march really runs on this slab (its class halos get built at the first smoother call)
                h.zero_vector(hi, "v")
                h.smooth(hi, mu)
                assert h.time_kernel("jacobik3", hi, 1) > 0
            h.prepare_cycle(hi)
            phase(h)
            h.zero_vector(hi, "v")
            res = h.vcycle(hi, 3, residuals=True)
            phase(h)
            full = h.get_vector(hi, "v", gather=True)
            h.set_params(mu, mu, 2 / 3, restriction="full_weighting")      # (drops the graphs: before the line-up)
            phase(h)
            h.zero_vector(hi, "v")
            h.vcycle(hi, 1)
            phase(h)
            fw = h.get_vector(hi, "v", gather=True)
            h.set_params(mu, mu, 1.0, smoother="rbgs")      # (drops the graphs: before the line-up)
            phase(h)
            h.zero_vector(hi, "v")
            h.vcycle(hi, 1)
            phase(h)
            gs = h.get_vector(hi, "v", gather=True)
            nrm = h.norm2(hi, "v")
            if tune.get("graph_comm"):                               # the cycles really were captured and replayed
                assert h.counters()["graph_replays"] >= 2, h.counters()
            results[rank] = (full, res, fw, gs, nrm)
            phase(h)
            h.close()
        except Exception as exc:                                     # noqa: BLE001
            import traceback
            traceback.print_exc()
            errors.append(exc)
            bar.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in threads), "a rank is stuck: the exchange pattern deadlocked"
    assert not errors, errors
    if "graph" in os.path.basename(os.environ["MG_RCCL_LIBRARY"]):
        fake = C.CDLL(os.environ["MG_RCCL_LIBRARY"])
        fake.fake_rccl_graph_timeouts.restype = C.c_longlong
        assert fake.fake_rccl_graph_timeouts() == 0, "a device-side handshake of the stand-in timed out"

    with DeviceHierarchy.synthetic(dim, lo, hi, c=c, mu1=mu, mu2=mu, **tune) as ser:
        ser.zero_vector(hi, "v")
        res = ser.vcycle(hi, 3, residuals=True)
        want = ser.get_vector(hi, "v")
        ser.set_params(mu, mu, 2 / 3, restriction="full_weighting")
        ser.zero_vector(hi, "v")
        ser.vcycle(hi, 1)
        want_fw = ser.get_vector(hi, "v")
        ser.set_params(mu, mu, 1.0, smoother="rbgs")
        ser.zero_vector(hi, "v")
        ser.vcycle(hi, 1)
        want_gs = ser.get_vector(hi, "v")
        want_nrm = ser.norm2(hi, "v")
    for rank in range(world):
        full, r, fw, gs, nrm = results[rank]
        assert np.array_equal(full, want), (rank, float(np.abs(full - want).max()))
        assert np.all(np.abs(r - res) <= 1e-13 * res)
        assert np.array_equal(fw, want_fw) and np.array_equal(gs, want_gs)
        assert abs(nrm - want_nrm) <= 1e-13 * want_nrm
    print(f"fake-rccl world={world} dim={dim} overlap={overlap}: OK")


def main_p2(world, dim, lo, hi, c, mu, replicate_below, overlap):
    """BASELINE config 5 on slabs: P2 lattice levels (two planes of reach -> halo_planes = 2), nine-colour
    Gauss-Seidel and weighted Jacobi, against a single-handle run bit for bit."""
    assert "fake_rccl" in os.environ.get("MG_RCCL_LIBRARY", "")
    buf = C.create_string_buffer(128)
    _capi.check(_capi.load().mg_comm_unique_id(buf, 128))
    uid = buf.raw
    results, errors = [None] * world, []
    tune = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("MG_TEST_TUNE", "").split(",") if kv}

    table = bool(os.environ.get("MG_TEST_P2_TABLE"))        # the P2 prolongation and its transpose (three fine planes of reach)
    extra = dict(transfers="p2", restriction="table") if table else {}

    def run(h):
        out = []
        for sm, om in (("mcgs", 1.0), ("jacobi", 0.6)):
            h.set_params(mu, mu, om, smoother=sm, restriction="table" if table else "direct")
            h.zero_vector(hi, "v")
            out.append(np.asarray(h.vcycle(hi, 2, residuals=True)))
            out.append(h.get_vector(hi, "v", gather=True))
        out.append(np.array([h.norm2(hi, "v")]))
        return out

    def rank_main(rank):
        try:
            def comm(h):
                h.set_tuning("overlap", overlap)
                h.set_tuning("overlap_min_rows", 0)
                h.set_comm_rccl(rank, world, uid, replicate_below=replicate_below)
            h = DeviceHierarchy.synthetic_p2(dim, lo, hi, c=c, mu1=mu, mu2=mu, comm=comm, halo_planes=2, **extra, **tune)
            info = h.level_info(hi)
            assert not info["replicated"] and info["n_local"] < info["n_global"]
            results[rank] = run(h)
            h.close()
        except Exception as exc:                                     # noqa: BLE001
            import traceback
            traceback.print_exc()
            errors.append(exc)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in threads), "a rank is stuck: the exchange pattern deadlocked"
    assert not errors, errors
    with DeviceHierarchy.synthetic_p2(dim, lo, hi, c=c, mu1=mu, mu2=mu, **extra, **tune) as ser:
        want = run(ser)
    for rank in range(world):
        got = results[rank]
        for k, (a, b) in enumerate(zip(got, want)):
            if a.ndim == 1:                                          # residual histories / norms: all-reduced sums
                assert np.all(np.abs(a - b) <= 1e-12 * np.abs(b)), (rank, k, a, b)
            else:
                assert np.array_equal(a, b), (rank, k, float(np.abs(a - b).max()))
    print(f"fake-rccl P2 world={world} dim={dim} overlap={overlap}: OK")


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:9]]
    if os.environ.get("MG_TEST_P2"):
        main_p2(*a)
    else:
        main(*a)
