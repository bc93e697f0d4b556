"""Worker functions for the multi-process tests (spawned with torch.multiprocessing)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def cpu_slab_worker(rank, world, port, dim, lo, hi, c, mu, replicate_below):
    """world ranks run the slab-decomposed CPU restatement over gloo; every rank checks it against the
    serial run of the same code (bit for bit) and against the oracle (round-off)."""
    from multigrid_dolfinx_amd import poisson
    from oracle.mg_oracle import Oracle
    from tests.dist_helpers import GlooTransport, SlabOracle, init_gloo
    dist = init_gloo(rank, world, port)
    try:
        bag = poisson.make_hierarchy(dim, lo, hi, c=c, mu1=mu, mu2=mu)
        par = SlabOracle(bag, dim, GlooTransport(dist, rank, world), rank, world, replicate_below)
        ser = SlabOracle(bag, dim, None, 0, 1, replicate_below)
        assert not par.lv[hi]["rep"] and par.lv[lo]["rep"]
        f = bag.b_dict[hi]
        vp, vs = par.new(hi), ser.new(hi)
        fp, fs = par.scatter(hi, f), ser.scatter(hi, f)
        orc = Oracle(bag, {l: L.grid_index for l, L in bag.levels.items()}, dim=dim)
        vo = np.zeros_like(f)
        for _ in range(2):
            vp = par.vcycle(hi, vp, fp)
            vs = ser.vcycle(hi, vs, fs)
            vo = orc.v_cycle(orc.A_jacobi_sp_dict[hi], vo, f)
            full = par.gather(hi, vp)
            assert np.array_equal(full, ser.owned(hi, vs)), "slab run differs from the serial run"
            assert np.linalg.norm(full - vo.ravel()) <= 1e-12 * np.linalg.norm(vo)
        # scalar all-reduce (norms)
        buf = np.array([float(np.sum(par.owned(hi, vp) ** 2))])
        par.t.allreduce(buf)
        assert abs(buf[0] - float(np.sum(ser.owned(hi, vs) ** 2))) <= 1e-12 * buf[0]
    finally:
        dist.destroy_process_group()


def rendezvous_worker(rank, world, port):
    """bench.py's bootstrap (id broadcast, barrier, max over ranks) on gloo."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import bench
    rv = bench.Rendezvous(world)
    try:
        payload = rv.broadcast_bytes(bytes(range(128)) if rank == 0 else None)
        assert payload == bytes(range(128))
        rv.barrier()
        assert rv.max(float(rank + 1)) == float(world)
    finally:
        rv.close()


def gpu_slab_worker(rank, world, port, dim, lo, hi, c, mu, replicate_below, mode, tuning=None):
    """world processes share GPU 0; slabs talk through the host-staged callback transport over gloo.
    Checks the distributed HIP path against a single-handle run on the same GPU."""
    from multigrid_dolfinx_amd import poisson
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    from tests.dist_helpers import GlooTransport, init_gloo
    dist = init_gloo(rank, world, port)
    try:
        t = GlooTransport(dist, rank, world)

        def comm(h):
            h.set_tuning("overlap_min_rows", 0)         # tiny grids: still take the overlapped sweep path
            h.set_comm_callbacks(rank, world, t.exchange, t.allreduce, t.allgatherv, replicate_below=replicate_below)

        bag = poisson.make_hierarchy(dim, lo, hi, c=c, mu1=mu, mu2=mu, seed=None if mode == "gen" else 3)
        gi = {l: L.grid_index for l, L in bag.levels.items()}
        f = bag.b_dict[hi]
        tuning = tuning or {}
        if mode == "gen":
            par = DeviceHierarchy.synthetic(dim, lo, hi, c=c, mu1=mu, mu2=mu, comm=comm, **tuning)
            ser = DeviceHierarchy.synthetic(dim, lo, hi, c=c, mu1=mu, mu2=mu, **tuning)
        else:
            par = DeviceHierarchy(dim, lo, hi, c=c, **tuning)
            comm(par)
            ser = DeviceHierarchy(dim, lo, hi, c=c, **tuning)
            for h in (par, ser):
                for l in range(lo, hi + 1):
                    A = bag.A_sp_dict[l][0]
                    if mode == "csr_local" and h is par:
                        # per-rank hand-off: only the rows of this rank's slab, in a local numbering of its own
                        # (owned nodes in reversed order, then the ghost nodes its rows couple to)
                        row0, nloc, _, _ = h.level_slab(l)
                        inv = np.empty_like(gi[l])
                        inv[gi[l]] = np.arange(gi[l].size)
                        A_lex = A.tocsr()[inv][:, inv].tocsr()
                        owned = np.arange(row0, row0 + nloc)[::-1]
                        rows = A_lex[owned]
                        used = np.unique(rows.indices)
                        ghosts = used[(used < row0) | (used >= row0 + nloc)]
                        col_nodes = np.concatenate([owned, ghosts])
                        h.set_level_local(l, rows[:, col_nodes].tocsr(), col_nodes, gi[l])
                    else:
                        h.set_level(l, A, gi[l])
                h.set_params(mu, mu, bag.omega)
                h.set_vector(hi, "f", f)
        info = par.level_info(hi)
        if "fuse_min_rows" in tuning:           # the smoother really pairs sweeps on this slab
            assert par.time_kernel("jacobi2", hi, 1) > 0.0
        assert not info["replicated"] and par.level_info(lo)["replicated"]
        assert info["n_local"] < info["n_global"]
        for h in (par, ser):
            h.zero_vector(hi, "v")
        rp = par.vcycle(hi, 2, residuals=True)
        rs = ser.vcycle(hi, 2, residuals=True)
        got = par.get_vector(hi, "v", gather=True)
        want = ser.get_vector(hi, "v")
        assert np.array_equal(got, want), float(np.abs(got - want).max())
        assert np.all(np.abs(rp - rs) <= 1e-13 * rs)
        # owned-rows-only fetch leaves the other slab's entries untouched
        part = par.get_vector(hi, "v")
        own = part.ravel() != 0.0
        assert 0 < own.sum() < part.size and np.array_equal(part.ravel()[own], want.ravel()[own])
        # full weighting needs fine halos of the residual; FMG needs coarse halos for prolongation
        for h in (par, ser):
            h.set_params(mu, mu, bag.omega, restriction="full_weighting")
            h.zero_vector(hi, "v")
            h.vcycle(hi, 1)
        assert np.array_equal(par.get_vector(hi, "v", gather=True), ser.get_vector(hi, "v"))
        for h in (par, ser):
            h.set_params(mu, mu, bag.omega)
            if mode != "gen":
                for l in range(lo, hi):
                    h.set_rhs_true(l, bag.b_dict[l])
            h.fmg(2)
        assert np.array_equal(par.get_vector(hi, "v", gather=True), ser.get_vector(hi, "v"))
        assert abs(par.norm2(hi, "v") - ser.norm2(hi, "v")) <= 1e-13 * ser.norm2(hi, "v")
        # red-black Gauss-Seidel: colours follow the GLOBAL index parity in every slab
        for h in (par, ser):
            h.set_params(mu, mu, 1.0, smoother="rbgs")
            h.zero_vector(hi, "v")
            h.vcycle(hi, 1)
        assert np.array_equal(par.get_vector(hi, "v", gather=True), ser.get_vector(hi, "v"))
        par.close()
        ser.close()
    finally:
        dist.destroy_process_group()
