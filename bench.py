#!/usr/bin/env python3
"""bench.py -- V-cycles/s and achieved HBM GB/s of the geometric-multigrid hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c4|c3|c2|c1] [--mu 50]

A "step" is one V(mu, mu) cycle (V_cycle_scheme, multigrid.py:231-268) of the synthetic P1
Poisson hierarchy named by --config, device-resident from start to end (the hierarchy and the
right-hand side are generated in HBM before the timed region).  Default = BASELINE.json's headline
workload: 3-D, 6 levels, N = 1024 elements per dimension (1025^3 unknowns), V(50,50), omega = 2/3 --
the reference's shipped smoother parameters (Multigrid_prototype.py:42-46).  With N > 1 GPUs the
same grid is split into slabs (strong scaling); either launch as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N ...
or plainly as `python bench.py --gpus N`, which starts exactly that as a child process (before touching a GPU)
and relays its line.  torch is used for that rendezvous only (RCCL id broadcast, barrier, max over ranks; gloo); the data
path is libmg_hip.so + RCCL.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for every field.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (dim, coarsest_level, finest_level, description)            N_l = 8 * 2**l
    "c1": (2, 1, 3, "2D Poisson P1, 3-level 64x64 fine grid"),
    "c2": (2, 4, 8, "2D Poisson P1, 5-level 2048x2048 fine grid"),
    "c3": (3, 2, 5, "3D Poisson P1, 4-level 256^3 fine grid"),
    "c4": (3, 2, 7, "3D Poisson P1, 6-level 1024^3 fine grid"),
    # BASELINE.json config 5 (no reference counterpart: the reference is P1 / Jacobi only): P2 elements on the
    # 513^3-point lattice (256 cells per dimension), nine-colour Gauss-Seidel, V(2,2) unless --mu says otherwise
    "c5": (3, 2, 6, "3D Poisson P2, 5-level 513^3-point lattice (512^3 steps), nine-colour Gauss-Seidel"),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def kernel_source_sha():
    """Identifies the kernel sources a PMC traffic figure belongs to (profiles/traffic.json)."""
    import hashlib
    hsh = hashlib.sha256()
    base = os.path.join(ROOT, "multigrid_dolfinx_amd", "csrc")
    for name in ("mg_kernels.hip.h", "mg_jacobi2.hip.h", "mg_jacobik3d.hip.h", "mg_jacobiblk.hip.h", "mg_lattice.hip.h", "mg_direct.hip.h", "mg_capi.hip"):
        try:
            hsh.update(open(os.path.join(base, name), "rb").read())
        except OSError:
            return None
    return hsh.hexdigest()[:16]


def headline_metric():
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "V-cycles/sec + achieved HBM GB/s, 3D Poisson P1 1024^3 DoF, 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--mu", type=int, default=None, help="pre- and post-smoothing sweeps (default: the reference's 50; "
                                                         "2 for the Gauss-Seidel configuration c5)")
    ap.add_argument("--omega", type=float, default=None, help="default 2/3 (Jacobi), 1 (Gauss-Seidel)")
    ap.add_argument("--rows-per-lane", type=int, default=None)
    ap.add_argument("--xcd-chunk", type=int, default=None)
    ap.add_argument("--offset-codes", type=int, default=None, help="0 = int32 column indices")
    ap.add_argument("--strip-slices", type=int, default=None, help="XCD strip traversal (0 = off)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="any other mg_set_tuning knob, e.g. --tune fuse_sweeps=0 (one Jacobi sweep per launch)")
    ap.add_argument("--nontemporal", type=int, default=None)
    ap.add_argument("--coarse-direct", type=int, default=None, help="0 = PCG on the coarsest level")
    ap.add_argument("--symmetric-storage", type=int, default=None, help="0 = keep lower entries and codes")
    ap.add_argument("--graph", type=int, default=None, help="0 = launch every kernel of a V-cycle eagerly")
    ap.add_argument("--lds-pad", type=int, default=None, help="experiment: dynamic LDS bytes per block")
    ap.add_argument("--replicate-below", type=int, default=1 << 22,
                    help="levels with fewer unknowns are replicated on every rank")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "gloo"],
                    help="slab transport for N > 1: RCCL over xGMI (default) or host-staged gloo callbacks "
                         "(debugging on a box with fewer GPUs than ranks; slow, never a headline number)")
    ap.add_argument("--kernel-reps", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-odd-rows", action="store_true", help="skip the extra hierarchy with 1 %% odd rows (value_with_odd_rows)")
    ap.add_argument("--cpu-sample-level", type=int, default=4, help="finest level of the CPU sample (N = 8*2^l)")
    return ap.parse_args()


class _StdoutToStderr:
    """gloo announces its connections on the C-level stdout; the contract is ONE JSON line on stdout, so file
    descriptor 1 points at stderr while torch.distributed sets itself up or tears itself down."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


class Rendezvous:
    """torch.distributed (gloo) for bootstrap, barrier and max-over-ranks only."""

    def __init__(self, gpus):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != gpus:
            raise SystemExit(f"--gpus {gpus} does not match WORLD_SIZE {self.world}")
        self.dist = None
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("NCCL_DEBUG", "WARN")        # RCCL problems show up on stderr
            with _StdoutToStderr():
                import torch.distributed as dist
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
                dist.barrier()              # forces the lazy pair connections (and their messages) now
            self.dist = dist

    def broadcast_bytes(self, payload):
        if self.dist is None:
            return payload
        box = [payload]
        self.dist.broadcast_object_list(box, src=0)
        return box[0]

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max(self, x):
        if self.dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def close(self):
        if self.dist is not None:
            with _StdoutToStderr():
                self.dist.destroy_process_group()


class GlooCallbacks:
    """Host-staged slab transport over gloo (mg_set_comm_callbacks); debugging only."""

    def __init__(self, rv):
        self.rv, self.dist = rv, rv.dist

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi):
        import torch
        reqs = []
        if recv_lo is not None:
            reqs.append(self.dist.irecv(torch.from_numpy(recv_lo), src=self.rv.rank - 1))
        if recv_hi is not None:
            reqs.append(self.dist.irecv(torch.from_numpy(recv_hi), src=self.rv.rank + 1))
        if send_lo is not None:
            reqs.append(self.dist.isend(torch.from_numpy(np.ascontiguousarray(send_lo)), dst=self.rv.rank - 1))
        if send_hi is not None:
            reqs.append(self.dist.isend(torch.from_numpy(np.ascontiguousarray(send_hi)), dst=self.rv.rank + 1))
        for r in reqs:
            r.wait()

    def allreduce(self, buf):
        import torch
        self.dist.all_reduce(torch.from_numpy(buf))

    def allgatherv(self, send, recv, counts):
        import torch
        off = 0
        for r in range(self.rv.world):
            seg = recv[off:off + int(counts[r])]
            if r == self.rv.rank:
                seg[:] = send
            self.dist.broadcast(torch.from_numpy(seg), src=r)
            off += int(counts[r])


def build_hierarchy(args, rv):
    from multigrid_dolfinx_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):          # a checkout without the (git-ignored) build product
        if rv.rank == 0:
            _capi.build_extension()
        rv.barrier()
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    import ctypes as C
    dim, lo, hi, _ = CONFIGS[args.config]
    tuning = {}
    if args.rows_per_lane:
        tuning["rows_per_lane"] = args.rows_per_lane
    if args.xcd_chunk:
        tuning["xcd_chunk"] = args.xcd_chunk
    if args.offset_codes is not None:
        tuning["offset_codes"] = args.offset_codes
    if args.strip_slices is not None:
        tuning["strip_slices"] = args.strip_slices
    if args.nontemporal is not None:
        tuning["nontemporal"] = args.nontemporal
    if args.coarse_direct is not None:
        tuning["coarse_direct"] = args.coarse_direct
    if args.symmetric_storage is not None:
        tuning["symmetric_storage"] = args.symmetric_storage
    if args.graph is not None:
        tuning["graph"] = args.graph
    if args.lds_pad is not None:
        tuning["lds_pad"] = args.lds_pad
    for kv in args.tune:
        key, value = kv.split("=")
        tuning[key] = int(value)

    if rv.world > 1 and args.transport == "rccl":
        # every rank must be able to load RCCL before anybody enters the collective communicator set-up;
        # if one cannot, ALL ranks switch to the host-staged transport (reported in config.parallelism)
        ok = 1.0
        try:
            _capi.check(_capi.load().mg_comm_unique_id(C.create_string_buffer(128), 128))
        except Exception as exc:            # noqa: BLE001 - any failure means "no RCCL here"
            print(f"rank {rv.rank}: RCCL unavailable ({exc}); falling back to gloo host staging", file=sys.stderr)
            ok = 0.0
        if rv.max(1.0 - ok) > 0.0:
            args.transport = "gloo"

    def comm(h):
        if rv.world == 1:
            return
        if args.transport == "gloo":
            t = GlooCallbacks(rv)
            h.set_comm_callbacks(rv.rank, rv.world, t.exchange, t.allreduce, t.allgatherv,
                                 replicate_below=args.replicate_below)
            return
        uid = None
        if rv.rank == 0:
            buf = C.create_string_buffer(128)
            _capi.check(_capi.load().mg_comm_unique_id(buf, 128))
            uid = buf.raw
        uid = rv.broadcast_bytes(uid)
        h.set_comm_rccl(rv.rank, rv.world, uid, replicate_below=args.replicate_below)

    device = rv.local_rank
    if args.transport == "gloo":            # ranks may share a GPU in this debugging mode
        import torch
        device = rv.local_rank % max(1, torch.cuda.device_count())
    if rv.world > 1 and args.config != "c5":
        # slabs: room for five halo planes, so that five sweeps share one grouped exchange (the K-sweep march on slabs)
        tuning.setdefault("halo_depth", 5)
    if args.config == "c5":
        if rv.world > 1:
            tuning["halo_planes"] = 2           # P2 rows reach two lattice planes
            tuning.setdefault("halo_depth", 3)  # ... and the transpose of the P2 prolongation three fine planes
        # transfers: the P2 prolongation and its transpose as restriction (the same algorithm on one GPU and on slabs)
        return DeviceHierarchy.synthetic_p2(dim, lo, hi, c=8, mu1=args.mu, mu2=args.mu, omega=args.omega, device=device,
                                            comm=comm, transfers="p2", restriction="table", **tuning)
    return DeviceHierarchy.synthetic(dim, lo, hi, c=8, mu1=args.mu, mu2=args.mu, omega=args.omega,
                                     prune_zeros=True, device=device, comm=comm, **tuning)


def timed_cycles(h, rv, level, warmup, steps):
    h.zero_vector(level, "v")
    if warmup:
        h.vcycle(level, warmup)
    h.sync()
    rv.barrier()
    t0 = time.perf_counter()
    h.vcycle(level, steps)
    h.sync()
    rv.barrier()
    return rv.max(time.perf_counter() - t0)


def cpu_baseline(args):
    """The oracle (a port of the reference's NumPy/SciPy V-cycle) on a bounded sample of the same
    workload: same dimension, smoother and level count rule, smaller fine grid; 1 thread, like the
    reference (SciPy csr_matvec / SuperLU do not thread)."""
    from multigrid_dolfinx_amd import poisson
    from oracle.mg_oracle import Oracle
    dim, lo, hi, _ = CONFIGS[args.config]
    p2 = args.config == "c5"
    s_hi = min(hi, (3 if p2 else args.cpu_sample_level) if dim == 3 else 7)     # same coarsest grid, fewer fine levels
    s_lo = min(lo, s_hi - 1)
    if p2:
        import types
        levels = {l: poisson.p2_level(4 * 2 ** l, dim) for l in range(s_lo, s_hi + 1)}
        bag = types.SimpleNamespace(
            mesh_dof_list_dict={}, element_size={l: 1.0 / L.N for l, L in levels.items()}, coarsest_level_elements_per_dim=8,
            coarsest_level=s_lo, finest_level=s_hi, A_sp_dict={l: (L.A, l) for l, L in levels.items()}, A_jacobi_sp_dict={},
            b_dict={l: L.b for l, L in levels.items()}, mu0=1, mu1=args.mu, mu2=args.mu, omega=args.omega,
            residual_per_V_cycle_finest=[], error_per_V_cycle_finest=[], u_exact_fine=None, V_fine_dolfx=None, levels=levels)
    else:
        bag = poisson.make_hierarchy(dim, s_lo, s_hi, c=8, mu0=1, mu1=args.mu, mu2=args.mu, omega=args.omega)
    orc = Oracle(bag, {l: L.grid_index for l, L in bag.levels.items()}, dim=dim)
    if p2:                                      # the same transfer pair as the device run
        orc.prolongation_table = poisson.p2_prolongation_table(dim)
        orc.restriction_table = poisson.p2_restriction_table(dim)
    f = bag.b_dict[s_hi]
    v = np.zeros_like(f)
    cycles, t0 = 0, time.perf_counter()
    while cycles < 1 or (time.perf_counter() - t0 < 10.0 and cycles < 20):
        v = orc.v_cycle(orc.A_jacobi_sp_dict[s_hi], v, f, smoother="mcgs" if p2 else "jacobi",
                        restriction="table" if p2 else "direct")
        cycles += 1
    dt = time.perf_counter() - t0
    n_s = bag.levels[s_hi].n
    n_full = (8 * 2 ** hi + 1) ** dim
    per_s = cycles / dt
    return {
        "value": per_s * n_s / n_full, "unit": "V-cycles/s", "cores": 1, "kind": "port",
        "sample": (f"oracle/mg_oracle.py V({args.mu},{args.mu}) on a {s_hi - s_lo + 1}-level {dim}-D hierarchy with "
                   f"N={8 * 2 ** s_hi} ({n_s} DoF): {cycles} cycles in {dt:.2f} s = {per_s:.4f} cycles/s on the "
                   f"sample, scaled by DoF ({n_s}/{n_full}) to the benchmarked grid; host has {os.cpu_count()} "
                   f"logical CPUs, 1 thread used"),
        "sample_cycles_per_s": per_s, "sample_dofs": n_s,
    }


def launch_own_ranks(gpus):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves -- as fresh child
    processes (`python -m torch.distributed.run ... bench.py <same arguments>`), BEFORE this process has made any GPU
    call (it never makes one) -- relay rank 0's JSON line and return the children's exit code.  Nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket() as sock:                   # a free rendezvous port on the loopback interface
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env)
    line = None
    for raw in proc.stdout:                         # the launcher and the ranks keep stdout for the one JSON line
        text = raw.decode(errors="replace")
        if text.lstrip().startswith("{") and '"metric"' in text:
            line = text
        else:
            sys.stderr.write(text)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    elif rc == 0:
        rc = 1
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_own_ranks(args.gpus))
    p2 = args.config == "c5"
    if args.mu is None:
        args.mu = 2 if p2 else 50
    if args.omega is None:
        args.omega = 1.0 if p2 else 2.0 / 3.0
    # Contract: ONE JSON line on stdout.  RCCL (version banner at communicator set-up), gloo and the HIP
    # runtime write to the C-level stdout at times, so file descriptor 1 points at stderr for the whole run and
    # the JSON line goes to the real stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rv = Rendezvous(args.gpus)
    dim, lo, hi, desc = CONFIGS[args.config]
    t_setup = time.perf_counter()
    h = build_hierarchy(args, rv)
    h.sync()
    t_setup = time.perf_counter() - t_setup

    elapsed = timed_cycles(h, rv, hi, args.warmup, args.steps)
    res_after = h.vcycle(hi, 1, residuals=True)[0]          # untimed: convergence evidence
    f_norm = h.norm2(hi, "f")

    # dominant kernel: the fine-level Jacobi launch, HIP events on the handle's stream
    info = h.level_info(hi)
    jac_ms = h.time_kernel("jacobi", hi, args.kernel_reps)
    res_ms = h.time_kernel("residual", hi, args.kernel_reps)
    gs_ms = h.time_kernel("gs", hi, max(2, args.kernel_reps // 4)) if p2 else None
    multi_k, small, march_k = 0, False, 0
    tuned = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.tune}
    try:        # the smoother runs K sweeps per pass on this level (mg_jacobik3d.hip.h): that launch is the dominant one
        pair_ms = h.time_kernel("jacobik3", hi, args.kernel_reps)
        multi_k = march_k = min(tuned.get("fuse_k", 5), 3 if info["n_local"] < tuned.get("fuse_k4_min_rows", 0) else
                                5 if info["n_local"] < tuned.get("fuse_k_small_rows", 1 << 22) else
                                4 if info["n_local"] < tuned.get("fuse_k5_min_rows", 1 << 26) else 5)
    except Exception:
        pair_ms = None
    pair2_ms = None
    try:        # ... and pairs of sweeps where that does not apply / for what is left over (mg_jacobi2.hip.h)
        pair2_ms = h.time_kernel("jacobi2", hi, args.kernel_reps)
    except Exception:
        pass
    if pair_ms is None:
        pair_ms = pair2_ms
    if pair_ms is None:
        try:    # levels of a few thousand rows: all mu sweeps of a smoother call in one launch (sdia_jacobi_small)
            pair_ms = h.time_kernel("jacobi_small", hi, args.kernel_reps)
            multi_k, small = args.mu, True
        except Exception:
            pair_ms = None
    if pair_ms is None:
        try:    # 2-D levels with row classes: up to five sweeps per launch (sdia_jacobik2d)
            pair_ms = h.time_kernel("jacobik", hi, args.kernel_reps)
            multi_k = next((int(kv.split("=")[1]) for kv in args.tune if kv.startswith("fuse_2d_k=")), 5)
        except Exception:
            pair_ms = None
    user_tuning = bool(args.tune) or any(v is not None for v in (
        args.rows_per_lane, args.xcd_chunk, args.offset_codes, args.strip_slices, args.nontemporal,
        args.symmetric_storage, args.lds_pad))
    has_classes = info.get("row_classes", 0) > 0
    classes_in_pair = has_classes and not any(kv.startswith("fuse_classes=0") for kv in args.tune)
    classes_in_sweep = has_classes and not any(kv.startswith("class_sweeps=0") for kv in args.tune)
    n_loc, z_loc = info["n_local"], info["nnz_nonzero"]
    W = info["ell_width"]

    def format_bytes_per_row(classes):
        """Bytes the shipped storage format MUST stream per row and launch: the matrix as stored
        + read x, read f, write out (+ D^-1 where it is streamed); one pass also when the launch does two sweeps."""
        if classes:                                     # one class byte per row, whatever format holds the rows themselves
            return 25
        if info["symmetric_diagonals"]:
            return 8 * info["symmetric_diagonals"] + 24
        if info["offset_codes"]:
            return 8 * W + 8 * ((W + 7) // 8) + 24
        return 12 * W + 32

    sweeps_per_launch = (multi_k or 2) if pair_ms else 1
    dom_ms = pair_ms if pair_ms else (gs_ms if gs_ms else jac_ms)
    dom_classes = classes_in_pair if (pair_ms and not gs_ms) else classes_in_sweep
    fmt_row = format_bytes_per_row(dom_classes)
    bytes_launch = fmt_row * n_loc
    achieved = bytes_launch / (dom_ms * 1e-3) / 1e9
    # SURVEY.md 8(d)'s CSR byte model of the same work, for comparison only (never a roofline fraction: the shipped
    # formats do not move these bytes)
    csr_model_bytes = sweeps_per_launch * (12 * z_loc + 36 * n_loc)
    if gs_ms:
        # 3-D levels with stencil classes run the colour launches as a plane march with x in LDS (mg_lattice.hip.h)
        march = has_classes and dim == 3 and "lattice_march=0" not in args.tune
        pairs = march and args.gpus == 1 and "lattice_gs2=1" in args.tune            # (opt-in: measured slower)
        kernel_id = ("lat_gs2 x 5 launches (two colours each)" if pairs else "lat_march<MODE_GS> x 9 colours" if march else
                     "ell_cls_apply<2, MODE_GS> x 9 colours" if has_classes else "ell_apply_coded<0, 2, MODE_GS> x 9 colours")
    elif small:
        need = (info["n_local"] + 1023) // 1024         # rows per thread: the smallest instance that covers the level
        kernel_id = "sdia_jacobi_small<%d, %d>" % (3 if dim == 2 else 4, next(r for r in (1, 2, 3, 4, 5, 6, 8, 12, 16) if r >= need))
    elif march_k:
        shape = {0: "12, 2, 2", 1: "12, 4, 1", 2: "8, 3, 2", 3: "6, 4, 1", 4: "8, 3, 1", 5: "4, 6, 1", 6: "16, 3, 1", 7: "16, 2, 1"}[tuned.get("fuse_k_shape", 7)]
        kernel_id = f"sdia_jacobikc_finest<{march_k}, {shape}>"
    elif multi_k:
        # lines per region as mg_capi.hip chooses them: 16 while all tiles of the level run at once (two workgroups per CU), else 32
        side = int(round(info["n_global"] ** 0.5))
        tiles16 = -(-side // (128 - 2 * multi_k)) * -(-side // (16 - 2 * multi_k))
        lines = tuned.get("fuse_2d_lines", 0) or (16 if tiles16 <= 2 * 256 else 32)
        kernel_id = f"sdia_jacobik2d<{multi_k}, {lines}>"       # (the finest level of the 2-D configurations takes 64-line regions)
    elif pair_ms:
        # (without row classes: the round-2 structure of the pass on the stored rows, unless "fuse_plain" 1 asks for round 1's)
        plain = {0: "sdia_jacobi2p_finest<2, 12, 1>", 1: "sdia_jacobi2p_finest<2, 8, 2>", 2: "sdia_jacobi2p_finest<2, 16, 1>"}[
            tuned.get("fuse_plain_shape", 0)] if tuned.get("fuse_plain", 2) == 2 else "sdia_jacobi2_finest<2, 8, 2, false>"
        kernel_id = "sdia_jacobi2c_finest<12, 2>" if dom_classes else plain
    elif info["symmetric_diagonals"]:
        kernel_id = ("sdia_cls_jacobi_finest<%d, 2, true>" % info["symmetric_diagonals"]) if dom_classes else \
                    ("sdia_jacobi_finest<%d, 2, true>" % info["symmetric_diagonals"])
    else:
        kernel_id = ("ell_apply_coded" if info["offset_codes"] else "ell_apply") + "<..., MODE_JACOBI>"
    # measured HBM-side traffic of exactly this kernel (PMC passes, profiles/): only quoted when the figure was
    # taken from the kernel sources that are running now, on this configuration, with default tuning
    traffic, traffic_note = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    src_sha = kernel_source_sha()
    if os.path.exists(tpath):
        try:
            entry = json.load(open(tpath)).get("entries", {}).get(f"{args.config}_gpus{args.gpus}:{kernel_id}")
            if entry is None:
                traffic_note = "no PMC figure for this kernel / configuration"
            elif user_tuning:
                traffic_note = "tuning differs from the profiled run"
            elif entry.get("kernel_source_sha") != src_sha:
                traffic_note = f"kernel sources changed since {entry.get('profile')} was taken"
            else:
                traffic = entry["hbm_bytes_per_launch"]
                traffic_note = f"{entry.get('profile')}: FETCH_SIZE x 2 + WRITE_SIZE at the L2 / fabric boundary (Infinity Cache hits included): an upper bound on the HBM bytes"
        except Exception as exc:           # noqa: BLE001
            traffic_note = f"profiles/traffic.json unreadable: {exc}"

    # the rate a level WITHOUT row classes gets (more than 255 distinct rows: variable coefficients, row-dependent
    # round-off in the assembly): the same cycles with the class byte switched off, timed the same way
    value_plain = pair_plain_ms = None
    if has_classes and not user_tuning:
        h.set_tuning("fuse_classes", 0)
        h.set_tuning("class_sweeps", 0)
        try:
            k_plain = max(1, min(args.steps, 5))
            value_plain = k_plain / timed_cycles(h, rv, hi, 1, k_plain)
            if pair_ms and (not multi_k or march_k):
                pair_plain_ms = h.time_kernel("jacobi2", hi, max(2, args.kernel_reps // 2))
        finally:
            h.set_tuning("fuse_classes", 1)
            h.set_tuning("class_sweeps", 1)

    # conventional V(2,2) for information
    h.set_params(2, 2, args.omega, smoother="mcgs" if p2 else "jacobi",
                 restriction="table" if p2 else "direct")
    v22 = timed_cycles(h, rv, hi, 1, max(2, args.steps))
    v22_per_s = max(2, args.steps) / v22
    mem = h.memory_bytes()
    dev = h.device_info()
    h.close()

    # ... and the rate a level with a FEW rows unlike any other gets ("row_escape": the frequent rows keep their classes, the
    # odd ones are fetched from the stored matrix inside the K-sweep pass): the same hierarchy generated with 1 % of the
    # interior rows carrying a reaction term of their own ("gen_odd_rows"), one rank, the headline configuration only
    odd_rows = None
    if args.config == "c4" and args.gpus == 1 and has_classes and not user_tuning and not args.no_odd_rows:
        from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
        with DeviceHierarchy.synthetic(dim, lo, hi, c=8, mu1=args.mu, mu2=args.mu, omega=args.omega, gen_odd_rows=100) as ho:
            st = ho.level_storage(hi)
            k_odd = max(1, min(args.steps, 3))
            odd_rows = {"share": 0.01, "escape_rows": st["escape_rows"], "row_classes": ho.level_info(hi)["row_classes"],
                        "value": k_odd / timed_cycles(ho, rv, hi, 1, k_odd),
                        "note": "same V-cycles on a hierarchy whose levels have 1 % rows unlike any other (more than 255 distinct rows)"}

    out = None
    if rv.rank == 0:
        per_s = args.steps / elapsed
        out = {
            "metric": headline_metric() if args.config == "c4" else f"V-cycles/sec + achieved HBM GB/s, {desc}",
            "value": per_s, "unit": "V-cycles/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{desc}, {(8 * 2 ** hi + 1) ** dim} DoF, V({args.mu},{args.mu}) omega={args.omega:.4f}, "
                                   f"P2 prolongation and its transpose as restriction, exact block-LU coarsest solve, up to 51 "
                                   f"entries per row; "
                                   f"no reference implementation exists for this configuration (parity unpinned)" if p2 else
                                   f"{desc}, {(8 * 2 ** hi + 1) ** dim} DoF, V({args.mu},{args.mu}) weighted Jacobi "
                                   f"omega={args.omega:.4f}, injection, Q1 prolongation, exact block-LU coarsest solve, "
                                   f"explicit zeros pruned (7-point rows)" if dim == 3 else
                                   f"{desc}, {(8 * 2 ** hi + 1) ** dim} DoF, V({args.mu},{args.mu}) weighted Jacobi",
                       "levels": hi - lo + 1, "dim": dim, "elements_per_dim": 8 * 2 ** hi,
                       "parallelism": (f"slab{args.gpus}" + ("-gloo-host-staged" if args.transport == "gloo" else ""))
                                      if args.gpus > 1 else "single"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": kernel_id + (" (one fine-level nine-colour Gauss-Seidel sweep; kernel_ms is the sum of its launches"
                                                if gs_ms else
                                                f" ({sweeps_per_launch} fine-level weighted-Jacobi sweeps per launch" if pair_ms else
                                                " (one fine-level weighted-Jacobi sweep per launch") +
                                   (", class-coded rows)" if dom_classes else ")"),
                         "bytes_model": (f"{fmt_row} B per row and launch = what the shipped format must stream: "
                                         + ("1 class byte" if dom_classes else "matrix as stored")
                                         + " + x 8 + f 8 + out 8" + ("" if info["symmetric_diagonals"] or info["offset_codes"] else " + D^-1 8")
                                         + "; rows_per_launch x that / kernel_ms"),
                         "format_bytes_per_row": fmt_row, "sweeps_per_launch": sweeps_per_launch,
                         "kernel_ms": dom_ms, "algorithmic_bytes_per_launch": bytes_launch,
                         "rows_per_launch": n_loc, "nonzeros_per_launch": z_loc, "per_gpu": True,
                         "row_classes": info.get("row_classes", 0),
                         "storage": ("symmetric diagonals" if info["symmetric_diagonals"] else
                                     "offset-coded ELL" if info["offset_codes"] else "ELL with int32 columns"),
                         # what crosses the L2/fabric boundary per second (PMC): includes tile rims and re-reads
                         "traffic_GBs": (traffic / (dom_ms * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac_of_peak": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "pair_pass_kernel_ms": pair2_ms,
                         "pair_pass_frac": (fmt_row * n_loc / (pair2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pair2_ms else None,
                         "single_sweep_kernel_ms": jac_ms,
                         "single_sweep_frac": format_bytes_per_row(classes_in_sweep) * n_loc / (jac_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "residual_kernel_ms": res_ms,
                         "residual_frac": (format_bytes_per_row(classes_in_sweep) - 0) * n_loc / (res_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "kernel_ms_without_row_classes": pair_plain_ms,
                         "frac_without_row_classes": (format_bytes_per_row(False) * n_loc / (pair_plain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                                                     if pair_plain_ms else None,
                         # SURVEY.md 8(d)'s CSR model of the same sweeps (12 z + 36 n per sweep) over the same time: how much
                         # faster than a kernel that streams CSR at the same byte rate -- NOT a fraction of the roofline
                         "csr_model_bytes_per_launch": csr_model_bytes,
                         "speedup_vs_csr_model": csr_model_bytes / bytes_launch},
            "value_without_row_classes": value_plain,
            "value_with_odd_rows": odd_rows,
            "v22_cycles_per_s": v22_per_s, "residual_l2_after": res_after, "rhs_l2": f_norm,
            "setup_s": t_setup, "device_memory_GB_per_gpu": mem / 1e9, "device": dev,
        }
        if not args.no_cpu_baseline and args.gpus == 1:
            out["cpu_baseline"] = cpu_baseline(args)
    rv.barrier()
    rv.close()
    sys.stdout.flush()
    if out is not None:
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    os.close(real_stdout)


if __name__ == "__main__":
    main()
