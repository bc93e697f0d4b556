"""The N>1 path.  CPU (gloo, world_size 2): the slab scheme itself -- aligned plane splits, one halo plane
per neighbour, replicated coarse levels -- restated in NumPy and checked against the serial oracle, plus
bench.py's rendezvous.  GPU: two processes sharing one MI355X run the real HIP kernels on two slabs through
the host-staged callback transport and must reproduce the single-handle result bit for bit."""
import pytest

from tests.dist_helpers import free_port, slab_splits


def _spawn(fn, world, *args):
    import torch.multiprocessing as mp
    mp.spawn(fn, args=(world, free_port()) + args, nprocs=world, join=True)


def test_slab_splits_are_aligned_between_levels():
    for N0, world in ((32, 8), (16, 2), (8, 8), (12, 5)):
        for l in range(0, 4):
            s, sf = slab_splits(N0, l, world), slab_splits(N0, l + 1, world)
            assert s[0] == 0 and s[-1] == N0 * 2 ** l + 1 and all(b > a for a, b in zip(s, s[1:]))
            for r in range(world):
                for K in range(s[r], s[r + 1]):
                    assert sf[r] <= 2 * K < sf[r + 1]              # coarse plane K lives with fine plane 2K


@pytest.mark.parametrize("world,dim,lo,hi,c,rep", [(2, 2, 1, 3, 8, 1000), (2, 3, 1, 3, 2, 200), (3, 2, 1, 3, 8, 1000)])
def test_slab_decomposition_reproduces_serial_cpu(world, dim, lo, hi, c, rep):
    from tests.dist_workers import cpu_slab_worker
    _spawn(cpu_slab_worker, world, dim, lo, hi, c, 3, rep)


def test_bench_rendezvous_over_gloo():
    from tests.dist_workers import rendezvous_worker
    _spawn(rendezvous_worker, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("world,dim,lo,hi,c,rep,mode", [(2, 2, 1, 3, 8, 1000, "csr"), (2, 3, 1, 3, 2, 200, "gen"),
                                                        (2, 2, 1, 3, 8, 1000, "csr_local"), (3, 3, 1, 3, 4, 200, "csr_local"),
                                                         (2, 3, 1, 3, 4, 0, "gen"), (4, 3, 1, 3, 4, 0, "gen"),
                                                         (3, 2, 1, 3, 8, 1000, "csr")])
def test_slabs_on_one_gpu_match_single_handle(world, dim, lo, hi, c, rep, mode):
    """2-4 processes share the GPU (interior ranks have two neighbours); results must equal the
    single-handle run bit for bit."""
    from tests.dist_workers import gpu_slab_worker
    _spawn(gpu_slab_worker, world, dim, lo, hi, c, 2, rep, mode)


@pytest.mark.gpu
def test_slab_interior_is_strip_walked_while_the_halo_travels():
    """129^3 unknowns on two slabs with 8-slice strips: the interior sub-range of every sweep uses the XCD
    strip traversal (a different slice order), the boundary slices the chunked map."""
    from tests.dist_workers import gpu_slab_worker
    _spawn(gpu_slab_worker, 2, 3, 2, 4, 8, 2, 0, "gen", {"strip_slices": 8})


@pytest.mark.gpu
@pytest.mark.parametrize("world,mu,segments", [(2, 2, 0), (3, 3, 0), (4, 5, 0), (2, 4, 4), (3, 5, 5)])
def test_paired_sweeps_on_slabs_match_single_handle(world, mu, segments):
    """The two-sweep kernel on slabs (129^3 and 65^3 unknowns, both distributed): interior rows get both sweeps in
    one pass, the slices holding a slab's first / last plane get their second sweep from the one-sweep kernel after
    the once-relaxed boundary planes have been exchanged.  Bit-identical to the single-handle run, odd sweep
    counts included.  With >= 4 plane segments per slab the pass is split so that both exchanges of a pair travel
    behind interior segments (boundary segments first)."""
    from tests.dist_workers import gpu_slab_worker
    _spawn(gpu_slab_worker, world, 3, 2, 4, 8, mu, 0, "gen", {"fuse_min_rows": 0, "fuse_segments": segments})


@pytest.mark.gpu
def test_rccl_entry_points_on_one_rank():
    """Every RCCL call the slab transport makes (unique id by value, init, all-reduce, grouped
    broadcast, grouped send/recv, destroy), on a one-rank communicator -- all a 1-GPU box can run."""
    from multigrid_dolfinx_amd import _capi
    _capi.check(_capi.load().mg_comm_selftest(0))


@pytest.mark.gpu
@pytest.mark.parametrize("world,dim,lo,hi,c,rep,overlap,tune", [
    (2, 3, 1, 3, 4, 0, 1, ""), (4, 3, 1, 3, 4, 0, 1, ""), (3, 2, 1, 3, 8, 1000, 1, ""), (4, 3, 1, 3, 4, 0, 0, ""),
    (2, 3, 2, 4, 8, 300000, 1, ""), (2, 3, 2, 4, 8, 0, 1, "fuse_min_rows=0"), (4, 3, 2, 4, 8, 300000, 0, "fuse_min_rows=0"),
    (2, 3, 2, 4, 8, 0, 1, "fuse_min_rows=0,fuse_segments=4"), (4, 3, 2, 4, 8, 300000, 1, "fuse_min_rows=0,fuse_segments=4")])
def test_asynchronous_rccl_code_path_with_in_process_stand_in(world, dim, lo, hi, c, rep, overlap, tune):
    """The library's RCCL branch (not the callback transport): grouped send/recv on the communication stream,
    ordered by events against the boundary / interior sweeps, all-reduce, grouped broadcasts -- executed against
    an in-process stand-in for librccl (tests/fake_rccl) with one thread per rank on one GPU.  A real multi-GPU
    RCCL run is not possible on the builder's box; this pins the call pattern and the stream ordering."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(here, "fake_rccl")], check=True)
    env = dict(os.environ, MG_RCCL_LIBRARY=lib, MG_TEST_TUNE=tune)
    out = subprocess.run([sys.executable, os.path.join(here, "fake_rccl_worker.py"), str(world), str(dim), str(lo),
                          str(hi), str(c), "2", str(rep), str(overlap)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "OK" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world,dim,lo,hi,c,rep,overlap,tune", [
    (2, 3, 1, 3, 4, 0, 1, "graph_comm=0"), (2, 3, 1, 3, 4, 0, 1, "graph_comm=1"), (4, 3, 1, 3, 4, 0, 0, "graph_comm=1"),
    (3, 2, 1, 3, 8, 1000, 1, "graph_comm=1"), (2, 3, 2, 4, 8, 300000, 1, "graph_comm=1,fuse_min_rows=0"),
    (4, 3, 2, 4, 8, 300000, 1, "graph_comm=1,fuse_min_rows=0,fuse_segments=4")])
def test_slab_cycles_captured_into_graphs_with_their_exchanges(world, dim, lo, hi, c, rep, overlap, tune):
    """`graph_comm`: a slab V-cycle -- sweeps on two streams, grouped send/recv, the grouped broadcasts of the replicated
    levels -- is captured into one hipGraph per rank and replayed.  RCCL's calls are capturable stream operations; the
    stand-in used here (tests/fake_rccl/fake_rccl_graph.hip) makes every send/recv a device-side handshake of kernel
    launches for the same reason.  Results equal the single-handle run bit for bit, the replay counter proves the graphs
    ran, and no handshake timed out.  (Real RCCL capture needs a multi-GPU node: the key stays opt-in.)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, "fake_rccl", "libfake_rccl_graph.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(here, "fake_rccl")], check=True)
    # every rank's streams must be able to run side by side: a spinning receive may not share a hardware queue with the
    # send it waits for
    env = dict(os.environ, MG_RCCL_LIBRARY=lib, MG_TEST_TUNE=tune, GPU_MAX_HW_QUEUES="24")
    out = subprocess.run([sys.executable, os.path.join(here, "fake_rccl_worker.py"), str(world), str(dim), str(lo),
                          str(hi), str(c), "2", str(rep), str(overlap)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "OK" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world,dim,lo,hi,c,rep,overlap,tune", [
    (2, 3, 1, 3, 4, 0, 1, ""), (3, 3, 1, 3, 4, 0, 0, ""), (2, 2, 1, 4, 8, 0, 1, ""), (4, 3, 1, 3, 4, 2000, 1, ""),
    (2, 3, 1, 3, 4, 0, 1, "lattice_march_min_rows=0"), (3, 3, 1, 3, 8, 0, 0, "lattice_march_min_rows=0"),
    (4, 3, 1, 3, 8, 2000, 1, "lattice_march_min_rows=0,lattice_segments=2"),
    (2, 3, 1, 3, 4, 0, 1, "halo_depth=3,table=1"), (3, 3, 1, 3, 8, 0, 0, "halo_depth=3,table=1,lattice_march_min_rows=0"),
    (2, 2, 1, 4, 8, 0, 1, "halo_depth=3,table=1")])
def test_p2_levels_on_slabs_with_two_plane_halos(world, dim, lo, hi, c, rep, overlap, tune):
    """BASELINE config 5 distributed: P2 rows reach two lattice planes, so the slabs exchange two planes per neighbour
    (`halo_planes` = 2).  Nine-colour Gauss-Seidel (a halo refresh after every colour) and weighted-Jacobi cycles on
    2-4 slabs -- threads over the in-process RCCL stand-in -- equal the single-handle run bit for bit; with
    `lattice_march_min_rows=0` the sweeps of these small slabs run as plane marches (mg_lattice.hip.h), whose tiles then
    read the neighbours' two halo planes."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(here, "fake_rccl")], check=True)
    env = dict(os.environ, MG_RCCL_LIBRARY=lib, MG_TEST_P2="1", MG_TEST_TUNE=",".join(kv for kv in tune.split(",") if kv != "table=1"))
    if "table=1" in tune:       # BASELINE config 5's own transfer pair on slabs: the restriction reaches three fine planes (halo_depth 3)
        env["MG_TEST_P2_TABLE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(here, "fake_rccl_worker.py"), str(world), str(dim), str(lo),
                          str(hi), str(c), "2", str(rep), str(overlap)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "OK" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world,mu,depth,fuse_k,overlap", [(2, 7, 4, 4, 1), (4, 9, 4, 4, 1), (3, 5, 3, 3, 0), (2, 11, 5, 5, 1), (4, 4, 2, 4, 1),
                                                          (3, 6, 4, 4, 1)])
def test_k_sweep_passes_on_slabs_match_single_handle(world, mu, depth, fuse_k, overlap):
    """`halo_depth` = K: the K-sweep march on slabs (129^3 and 65^3 unknowns, both distributed).  K planes of the iterate
    travel once per K sweeps, the slab relaxes its neighbours' K - 1 planes next to it itself -- through their row classes,
    translated into its own dictionary at set-up --, so there is no boundary chain.  2-4 ranks as threads over the
    in-process RCCL stand-in (asynchronous, event-ordered copies), with the boundary-planes-first overlap and without;
    sweep counts that split into passes of different sizes (7 = 4 + 3, 9 = 4 + 3 + 2, 11 = 5 + 4 + 2 ...), `halo_depth` 2
    (pairs only).  Bit-identical to the single-handle run."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    lib = os.path.join(here, "fake_rccl", "libfake_rccl.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(here, "fake_rccl")], check=True)
    tune = f"halo_depth={depth},fuse_k={fuse_k},fuse_min_rows=0,fuse_k_slab_min_rows=0,fuse_k_slab_min_sweeps=2"
    env = dict(os.environ, MG_RCCL_LIBRARY=lib, MG_TEST_TUNE=tune, MG_TEST_EXPECT_KSLAB="1")
    out = subprocess.run([sys.executable, os.path.join(here, "fake_rccl_worker.py"), str(world), "3", "2", "4", "8", str(mu), "0",
                          str(overlap)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "OK" in out.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world,mu,depth", [(2, 7, 4), (3, 5, 3)])
def test_k_sweep_passes_on_slabs_over_the_host_staged_transport(world, mu, depth):
    """The same through the callback transport, one PROCESS per rank sharing the GPU (edge and interior ranks, the class
    translation and every exchange over gloo): bit-identical to the single handle."""
    from tests.dist_workers import gpu_slab_worker
    _spawn(gpu_slab_worker, world, 3, 2, 4, 8, mu, 0, "gen", {"halo_depth": depth, "fuse_min_rows": 0, "fuse_k_slab_min_rows": 0, "fuse_k_slab_min_sweeps": 2})
