"""Device-resident multigrid hierarchy: a thin object wrapper over the C ABI.

The reference keeps its hierarchy in sixteen module globals (`multigrid.py:10-45`);
here it is one explicit handle per hierarchy.  All arithmetic happens in
`libmg_hip.so`; this module only marshals NumPy / SciPy buffers across ctypes.
Levels are addressed by the reference's integer level keys
(`coarsest_level .. finest_level`, `N_l = c * 2**l`).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import scipy.sparse as sp

from . import _capi
from ._capi import (MG_RESTRICT_FULL_WEIGHTING, MG_RESTRICT_INJECTION, MG_VEC_ERR, MG_VEC_F, MG_VEC_R,
                    MG_VEC_V, check, load, ptr)
from .poisson import grid_index_from_coords

__all__ = ["DeviceHierarchy", "jacobi_split"]

_VEC = {"v": MG_VEC_V, "f": MG_VEC_F, "r": MG_VEC_R, "err": MG_VEC_ERR}
_RESTRICT = {"direct": MG_RESTRICT_INJECTION, "injection": MG_RESTRICT_INJECTION,
             "full_weighting": MG_RESTRICT_FULL_WEIGHTING, "table": _capi.MG_RESTRICT_TABLE}


def _csr_arrays(A):
    if not sp.isspmatrix_csr(A):
        raise TypeError("the stiffness matrix must be a scipy.sparse.csr_matrix (Multigrid_prototype.py:96)")
    if np.iscomplexobj(A.data):
        raise TypeError("complex scalars are not supported (real fp64 only)")
    data = np.ascontiguousarray(A.data, dtype=np.float64)
    indices = np.ascontiguousarray(A.indices, dtype=np.int32)
    if A.indptr.dtype == np.int64:
        indptr, is64 = np.ascontiguousarray(A.indptr), 1
    else:
        indptr, is64 = np.ascontiguousarray(A.indptr, dtype=np.int32), 0
    return indptr, is64, indices, data


def jacobi_split(A, device=0):
    """`getJacobiMatrices` arithmetic on the GPU (`multigrid.py:48-56`).

    Returns `(DinvR csr, Dinv dia)` with the structure SciPy gives the reference:
    explicit zeros and the diagonal dropped, each row's entries in reversed order
    (`csr_matmat`), int32 indices, `has_sorted_indices == False`.
    """
    lib = load()
    indptr, is64, indices, data = _csr_arrays(A)
    n, nnz = A.shape[0], data.size
    dinv = np.empty(n)
    scaled = np.empty(nnz)
    keep = np.empty(nnz, dtype=np.uint8)
    check(lib.mg_jacobi_split(device, n, nnz, ptr(indptr), is64, ptr(indices), ptr(data), ptr(dinv),
                              ptr(scaled), ptr(keep)))
    keep = keep.astype(bool)
    rows = np.repeat(np.arange(n), np.diff(indptr))[keep]
    counts = np.bincount(rows, minlength=n)
    new_ptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(counts, out=new_ptr[1:])
    kept = int(new_ptr[-1])
    rank = np.arange(kept) - new_ptr[:-1][rows]
    dest = new_ptr[1:][rows] - 1 - rank                      # reversed inside every row
    out_idx = np.empty(kept, dtype=np.int32)
    out_val = np.empty(kept)
    out_idx[dest] = indices[keep]
    out_val[dest] = scaled[keep]
    R = sp.csr_matrix((out_val, out_idx, new_ptr), shape=A.shape)
    R.has_sorted_indices = False
    return R, sp.diags(dinv, 0)


class DeviceHierarchy:
    """One multigrid hierarchy resident on one MI355X (or one slab of it per rank)."""

    def __init__(self, dim: int, coarsest_level: int, finest_level: int, c: int = 8, device: int = 0,
                 rows_per_lane: Optional[int] = None, xcd_chunk: Optional[int] = None,
                 offset_codes: Optional[int] = None, strip_slices: Optional[int] = None,
                 nontemporal: Optional[int] = None, coarse_direct: Optional[int] = None,
                 symmetric_storage: Optional[int] = None, graph: Optional[int] = None,
                 lds_pad: Optional[int] = None, **more_tuning):
        self._lib = load()
        self.dim = dim
        self.c = c
        self.coarsest_level = coarsest_level
        self.finest_level = finest_level
        self.device = device
        self._h = C.c_void_p()
        check(self._lib.mg_create(finest_level - coarsest_level + 1, dim, device, C.byref(self._h)))
        self._keepalive = []
        if rows_per_lane is not None:
            self.set_tuning("rows_per_lane", rows_per_lane)
        if xcd_chunk is not None:
            self.set_tuning("xcd_chunk", xcd_chunk)
        if offset_codes is not None:
            self.set_tuning("offset_codes", offset_codes)
        if strip_slices is not None:
            self.set_tuning("strip_slices", strip_slices)
        if nontemporal is not None:
            self.set_tuning("nontemporal", nontemporal)
        if coarse_direct is not None:
            self.set_tuning("coarse_direct", coarse_direct)
        if symmetric_storage is not None:
            self.set_tuning("symmetric_storage", symmetric_storage)
        if graph is not None:
            self.set_tuning("graph", graph)
        if lds_pad is not None:
            self.set_tuning("lds_pad", lds_pad)
        for key, value in more_tuning.items():        # any other mg_set_tuning key (fuse_sweeps, fuse_segments ...)
            self.set_tuning(key, value)

    # ---- life cycle ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.mg_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _idx(self, level: int) -> int:
        if level < self.coarsest_level or level > self.finest_level:
            raise KeyError(f"level {level} outside {self.coarsest_level}..{self.finest_level}")
        return level - self.coarsest_level

    def elements(self, level: int) -> int:
        return self.c * 2 ** level

    def n_dofs(self, level: int) -> int:
        if getattr(self, "_flat_n", None) is not None:
            return self._flat_n
        return (self.elements(level) + 1) ** self.dim

    def device_info(self) -> str:
        buf = C.create_string_buffer(256)
        check(self._lib.mg_device_info(self._h, buf, 256))
        return buf.value.decode()

    # ---- set-up ----------------------------------------------------------------------------
    def set_tuning(self, key: str, value: int):
        check(self._lib.mg_set_tuning(self._h, key.encode(), int(value)))

    def set_comm_rccl(self, rank: int, world: int, unique_id: bytes, replicate_below: int = 1 << 21):
        buf = C.create_string_buffer(unique_id, len(unique_id))
        check(self._lib.mg_set_comm(self._h, rank, world, buf, len(unique_id), int(replicate_below)))

    def set_comm_callbacks(self, rank: int, world: int, exchange, allreduce, allgatherv,
                           replicate_below: int = 1 << 21):
        """Host-staged transport (tests): three Python callables moving NumPy buffers."""
        def _ex(user, slo, shi, rlo, rhi, count):
            try:
                as_np = lambda p: None if not p else np.ctypeslib.as_array(p, shape=(count,))
                exchange(as_np(slo), as_np(shi), as_np(rlo), as_np(rhi))
                return 0
            except Exception:          # pragma: no cover - surfaced as an MgError by the library
                import traceback
                traceback.print_exc()
                return 1

        def _ar(user, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:          # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        def _ag(user, send, send_count, recv, counts):
            try:
                cnt = np.ctypeslib.as_array(counts, shape=(world,)).copy()
                allgatherv(np.ctypeslib.as_array(send, shape=(send_count,)),
                           np.ctypeslib.as_array(recv, shape=(int(cnt.sum()),)), cnt)
                return 0
            except Exception:          # pragma: no cover
                import traceback
                traceback.print_exc()
                return 1

        cbs = (_capi.EXCHANGE_FN(_ex), _capi.ALLREDUCE_FN(_ar), _capi.ALLGATHERV_FN(_ag))
        self._keepalive.append(cbs)
        check(self._lib.mg_set_comm_callbacks(self._h, rank, world, cbs[0], cbs[1], cbs[2], None,
                                              int(replicate_below)))

    def set_level(self, level: int, A, grid_index=None, prune_zeros: bool = True):
        """Hand over one level's stiffness CSR (`Multigrid_prototype.py:95-99`)."""
        indptr, is64, indices, data = _csr_arrays(A)
        n = A.shape[0]
        if n != self.n_dofs(level):
            raise ValueError(f"level {level}: matrix has {n} rows, expected {self.n_dofs(level)}")
        gi = None
        if grid_index is not None:
            gi = np.ascontiguousarray(grid_index, dtype=np.int64)
            if gi.size != n:
                raise ValueError("grid_index has the wrong length")
            if np.array_equal(gi, np.arange(n)):
                gi = None
        check(self._lib.mg_set_level_csr(self._h, self._idx(level), self.elements(level), n, data.size,
                                         ptr(indptr), is64, ptr(indices), ptr(data),
                                         ptr(gi) if gi is not None else None, 1 if prune_zeros else 0))

    def set_prolongation(self, kind: str = "q1"):
        """"q1": the reference's bilinear / trilinear interpolation (`Interpolation2D`); "p2": the natural embedding of
        the coarse P2 space (`poisson.p2_prolongation_table`), for hierarchies of P2 lattice levels."""
        if kind == "q1":
            check(self._lib.mg_set_prolongation_table(self._h, None, None, None))
            return
        if kind != "p2":
            raise ValueError("prolongation must be 'q1' or 'p2'")
        from .poisson import p2_prolongation_table
        cnt, off, w = p2_prolongation_table(self.dim)
        cnt = np.ascontiguousarray(cnt, dtype=np.int32)
        off = np.ascontiguousarray(off, dtype=np.int32)
        w = np.ascontiguousarray(w, dtype=np.float64)
        check(self._lib.mg_set_prolongation_table(self._h, ptr(cnt), ptr(off), ptr(w)))
        # ... and its transpose, selected with set_params(restriction="table")
        from .poisson import p2_restriction_table
        rc, ro, rw = p2_restriction_table(self.dim)
        rc = np.ascontiguousarray(rc, dtype=np.int32)
        ro = np.ascontiguousarray(ro, dtype=np.int32)
        rw = np.ascontiguousarray(rw, dtype=np.float64)
        check(self._lib.mg_set_restriction_table(self._h, int(ro.shape[1]), ptr(rc), ptr(ro), ptr(rw)))

    def level_slab(self, level: int):
        """`(row0, n_local, halo_lo, halo_hi)`: the lexicographic nodes this rank owns on `level` and how many nodes
        below / above them it may couple to (before the level is set)."""
        v = [C.c_int64() for _ in range(4)]
        check(self._lib.mg_level_slab(self._h, self._idx(level), self.elements(level), *[C.byref(x) for x in v]))
        return tuple(int(x.value) for x in v)

    def set_level_local(self, level: int, A_local, col_nodes, grid_index=None, prune_zeros: bool = True):
        """Per-rank hand-off (`mg_set_level_csr_local`): `A_local` holds this rank's owned rows, its columns are local
        ids and `col_nodes[id]` their global lexicographic nodes (the first `A_local.shape[0]` ids are the rows)."""
        indptr, is64, indices, data = _csr_arrays(A_local)
        cn = np.ascontiguousarray(col_nodes, dtype=np.int64)
        if cn.size != A_local.shape[1]:
            raise ValueError("col_nodes must name every local column")
        gi = None
        if grid_index is not None:
            gi = np.ascontiguousarray(grid_index, dtype=np.int64)
            if np.array_equal(gi, np.arange(gi.size)):
                gi = None
        check(self._lib.mg_set_level_csr_local(self._h, self._idx(level), self.elements(level), A_local.shape[0],
                                               A_local.shape[1], data.size, ptr(indptr), is64, ptr(indices), ptr(data),
                                               ptr(cn), ptr(gi) if gi is not None else None, 1 if prune_zeros else 0))

    def gen_poisson_level(self, level: int, prune_zeros: bool = True):
        """Device-side synthetic level (same tiles as `poisson.make_level` + `set_level`)."""
        check(self._lib.mg_gen_poisson_level(self._h, self._idx(level), self.elements(level),
                                             1 if prune_zeros else 0))

    def gen_p2_level(self, level: int):
        """Device-side synthetic P2 level on the lattice with `elements(level)` steps per dimension (= twice the
        cells; BASELINE config 5), from the per-class interior stencils of `poisson.p2_stencils`."""
        from .poisson import p2_stencils
        if not hasattr(self, "_p2_tables"):
            self._p2_tables = p2_stencils(self.dim)
        count, offsets, values, load = self._p2_tables
        steps = self.elements(level)
        h = 2.0 / steps                                 # cell size
        vals = np.ascontiguousarray(values * h ** (self.dim - 2))
        ld = np.ascontiguousarray(load * h ** self.dim)
        cnt = np.ascontiguousarray(count, dtype=np.int32)
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        check(self._lib.mg_gen_lattice_level(self._h, self._idx(level), steps, int(cnt.max()), ptr(cnt), ptr(off),
                                             ptr(vals), ptr(ld)))

    @classmethod
    def synthetic_p2(cls, dim: int, coarsest_level: int, finest_level: int, c: int = 8, mu1: int = 2, mu2: int = 2,
                     omega: float = 1.0, smoother: str = "mcgs", device: int = 0, comm=None, transfers: str = "q1",
                     restriction: str = "direct", **tuning):
        """Whole P2 hierarchy from the device generator (lattices of c * 2^level steps per dimension); on slabs
        (`comm`) the tuning must carry halo_planes=2: P2 rows reach two lattice planes.  `transfers="p2"` +
        `restriction="table"`: the P2 prolongation and its transpose instead of the reference's bilinear table / injection."""
        h = cls(dim, coarsest_level, finest_level, c=c, device=device, **tuning)
        if comm is not None:
            comm(h)
        for level in range(coarsest_level, finest_level + 1):
            h.gen_p2_level(level)
        h.set_params(mu1, mu2, omega, smoother=smoother, restriction=restriction)
        if transfers == "p2":
            h.set_prolongation("p2")
        return h

    def set_params(self, mu1: int, mu2: int, omega: float, restriction: str = "direct",
                   coarse_rtol: float = 1e-14, coarse_maxit: int = 20000, keep_err: bool = False,
                   smoother: str = "jacobi"):
        sm = {"jacobi": _capi.MG_SMOOTH_JACOBI, "rbgs": _capi.MG_SMOOTH_RBGS, "mcgs": _capi.MG_SMOOTH_MCGS}[smoother]
        check(self._lib.mg_set_params(self._h, int(mu1), int(mu2), float(omega), _RESTRICT[restriction],
                                      sm, float(coarse_rtol), int(coarse_maxit), 1 if keep_err else 0))

    @classmethod
    def from_bag(cls, bag, dim: int = 2, grid_index: Optional[Dict[int, np.ndarray]] = None,
                 prune_zeros: bool = True, device: int = 0, keep_err: bool = True, **tuning):
        """Build from the reference's 16-field bag (`Multigrid_prototype.py:15-32`).

        The level link is taken from `grid_index[level]` if given, else recovered from
        the coordinate dictionaries `mesh_dof_list_dict` (`Multigrid_prototype.py:68-74`).
        """
        h = cls(dim, bag.coarsest_level, bag.finest_level, c=bag.coarsest_level_elements_per_dim,
                device=device, **tuning)
        for level in range(bag.coarsest_level, bag.finest_level + 1):
            A = bag.A_sp_dict[level][0]
            if grid_index is not None and level in grid_index:
                gi = grid_index[level]
            else:
                d = bag.mesh_dof_list_dict[level]
                n = A.shape[0]
                coords = np.array([d[j] for j in range(n)], dtype=np.float64)
                gi = grid_index_from_coords(coords, h.elements(level), dim)
            h.set_level(level, A, gi, prune_zeros=prune_zeros)
        h.set_params(bag.mu1, bag.mu2, bag.omega, keep_err=keep_err)
        return h

    @classmethod
    def synthetic(cls, dim: int, coarsest_level: int, finest_level: int, c: int = 8, mu1: int = 50,
                  mu2: int = 50, omega: float = 2.0 / 3.0, prune_zeros: bool = True, device: int = 0,
                  comm=None, **tuning):
        """Whole hierarchy from the device generator (bench and full-size tests)."""
        h = cls(dim, coarsest_level, finest_level, c=c, device=device, **tuning)
        if comm is not None:
            comm(h)
        for level in range(coarsest_level, finest_level + 1):
            h.gen_poisson_level(level, prune_zeros=prune_zeros)
        h.set_params(mu1, mu2, omega)
        return h

    # ---- queries -------------------------------------------------------------------------------
    def level_info(self, level: int) -> dict:
        vals = [C.c_int64() for _ in range(5)]
        w, rep, codes = C.c_int(), C.c_int(), C.c_int()
        check(self._lib.mg_level_info(self._h, self._idx(level), *[C.byref(v) for v in vals], C.byref(w),
                                      C.byref(rep), C.byref(codes)))
        keys = ("n_global", "n_local", "row0", "nnz_stored", "nnz_nonzero")
        out = {k: int(v.value) for k, v in zip(keys, vals)}
        out["ell_width"] = int(w.value)
        out["replicated"] = bool(rep.value)
        out["offset_codes"] = max(0, int(codes.value))
        out["symmetric_diagonals"] = max(0, -int(codes.value))
        ncls = C.c_int()
        check(self._lib.mg_level_row_classes(self._h, self._idx(level), C.byref(ncls)))
        out["row_classes"] = int(ncls.value)
        return out

    def level_storage(self, level: int) -> dict:
        """How the level's matrix got its storage (`mg_level_storage`): `symmetric` 1 = bit for bit, 2 = within
        `ulps_used` units in the last place, 0 = not (then `first_asymmetric_row`, in lexicographic numbering, and
        `max_pair_ulps`, -1 for a pair with one half missing, say why), -1 = not tested; `distinct_rows` seen by the row
        dictionary (more than 255: no row classes, or -- `escape_rows` > 0 -- classes for the frequent rows and that many
        rows read from the stored matrix), -1 = not built."""
        sym, rows, used = C.c_int(), C.c_int(), C.c_int()
        first, spread, esc = C.c_int64(), C.c_int64(), C.c_int64()
        check(self._lib.mg_level_storage(self._h, self._idx(level), C.byref(sym), C.byref(first), C.byref(spread),
                                         C.byref(rows), C.byref(used), C.byref(esc)))
        return {"symmetric": int(sym.value), "first_asymmetric_row": int(first.value), "max_pair_ulps": int(spread.value),
                "distinct_rows": int(rows.value), "ulps_used": int(used.value), "escape_rows": int(esc.value)}

    def memory_bytes(self) -> int:
        b = C.c_int64()
        check(self._lib.mg_memory_bytes(self._h, C.byref(b)))
        return int(b.value)

    # ---- vectors ---------------------------------------------------------------------------------
    def set_vector(self, level: int, which: str, x):
        a = _capi.as_f64(x, self.n_dofs(level))
        check(self._lib.mg_set_vector(self._h, self._idx(level), _VEC[which], ptr(a)))

    def set_rhs_true(self, level: int, x):
        a = _capi.as_f64(x, self.n_dofs(level))
        check(self._lib.mg_set_rhs_true(self._h, self._idx(level), ptr(a)))

    def get_vector(self, level: int, which: str, gather: bool = False) -> np.ndarray:
        out = np.zeros(self.n_dofs(level))
        check(self._lib.mg_get_vector(self._h, self._idx(level), _VEC[which], ptr(out), 1 if gather else 0))
        return out.reshape(-1, 1)

    def zero_vector(self, level: int, which: str = "v"):
        check(self._lib.mg_zero_vector(self._h, self._idx(level), _VEC[which]))

    def copy_vector(self, level: int, dst: str, src: str):
        check(self._lib.mg_copy_vector(self._h, self._idx(level), _VEC[dst], _VEC[src]))

    # ---- the hot path ------------------------------------------------------------------------------
    def smooth(self, level: int, nw: int):
        check(self._lib.mg_smooth(self._h, self._idx(level), int(nw)))

    def smooth_split(self, level: int, nw: int):
        """Sweeps in the reference's split-operand form (matrix = D^-1 R, "err" = diagonal of D^-1)."""
        check(self._lib.mg_smooth_split(self._h, self._idx(level), int(nw)))

    def residual(self, level: int):
        check(self._lib.mg_residual(self._h, self._idx(level)))

    def restrict(self, level: int, kind: str = "direct"):
        check(self._lib.mg_restrict(self._h, self._idx(level), _RESTRICT[kind]))

    def prolong(self, level: int, add: bool = True):
        check(self._lib.mg_prolong(self._h, self._idx(level), 1 if add else 0))

    def prepare_cycle(self, level):
        """Build now what the first V-cycle from `level` would build lazily (mg_prepare_cycle)."""
        check(self._lib.mg_prepare_cycle(self._h, self._idx(level)))

    def coarse_solve(self):
        it, rel = C.c_int(), C.c_double()
        check(self._lib.mg_coarse_solve(self._h, C.byref(it), C.byref(rel)))
        return int(it.value), float(rel.value)

    def vcycle(self, level: Optional[int] = None, ncycles: int = 1, residuals: bool = False):
        level = self.finest_level if level is None else level
        hist = np.zeros(max(1, ncycles)) if residuals else None
        check(self._lib.mg_vcycle(self._h, self._idx(level), int(ncycles), ptr(hist) if residuals else None))
        return hist[:ncycles] if residuals else None

    def norm2(self, level: int, which: str = "r") -> float:
        out = C.c_double()
        check(self._lib.mg_norm2(self._h, self._idx(level), _VEC[which], C.byref(out)))
        return float(out.value)

    def quadratic_form(self, level: int, which: str = "v") -> float:
        """x^T A x with the level's matrix (mass-matrix norms)."""
        out = C.c_double()
        check(self._lib.mg_quadratic_form(self._h, self._idx(level), _VEC[which], C.byref(out)))
        return float(out.value)

    def set_flat_space(self, n: int):
        """A single matrix-free level of n unknowns (vector norms)."""
        check(self._lib.mg_set_level_grid(self._h, 0, 0, int(n), None))
        self._flat_n = int(n)

    def fmg(self, mu0: int, tol: float = 0.0, max_cycles: int = 10000, top_level: Optional[int] = None,
            norm: str = "l2", errors: bool = False):
        """FMG on levels coarsest..top_level; returns the residual norm after every top-level cycle -- the l2 norm,
        or with norm="mass" the reference's L2(Omega) norm sqrt(r^T M r) (set_mass) -- and with `errors` also the
        norm of iterate - exact solution (set_exact) as a second array.  Everything stays on the device."""
        top = self.finest_level if top_level is None else top_level
        hist = np.zeros(max(mu0, max_cycles if tol > 0 else mu0, 1))
        ehist = np.zeros_like(hist) if errors else None
        done = C.c_int()
        check(self._lib.mg_fmg_ex(self._h, self._idx(top), int(mu0), float(tol), int(max_cycles),
                                  {"l2": _capi.MG_NORM_L2, "mass": _capi.MG_NORM_MASS}[norm], ptr(hist),
                                  ptr(ehist) if errors else None, C.byref(done)))
        if errors:
            return hist[:done.value].copy(), ehist[:done.value].copy()
        return hist[:done.value].copy()

    def set_mass(self, level: int, M):
        """The P1 mass matrix of `level` (SciPy CSR, the level's DoF numbering) for the L2(Omega) norms of fmg."""
        indptr, is64, indices, data = _csr_arrays(M)
        check(self._lib.mg_set_mass_csr(self._h, self._idx(level), M.shape[0], data.size, ptr(indptr), is64,
                                        ptr(indices), ptr(data)))

    def set_exact(self, level: int, u):
        """Nodal values of the exact solution on `level` (error history of fmg)."""
        a = _capi.as_f64(u, self.n_dofs(level))
        check(self._lib.mg_set_exact(self._h, self._idx(level), ptr(a)))

    def counters(self) -> dict:
        """Whole-vector host <-> device copies, replayed / cached V-cycle graphs so far (tests)."""
        up, down, rep = C.c_int64(), C.c_int64(), C.c_int64()
        cached = C.c_int()
        check(self._lib.mg_counters(self._h, C.byref(up), C.byref(down), C.byref(rep), C.byref(cached)))
        return {"uploads": int(up.value), "downloads": int(down.value), "graph_replays": int(rep.value),
                "graphs_cached": int(cached.value)}

    def set_level_grid(self, level: int, grid_index=None):
        """Geometry + numbering only (transfer operators without a matrix)."""
        n = self.n_dofs(level)
        gi = None
        if grid_index is not None:
            gi = np.ascontiguousarray(grid_index, dtype=np.int64)
            if gi.size != n:
                raise ValueError("grid_index has the wrong length")
            if np.array_equal(gi, np.arange(n)):
                gi = None
        check(self._lib.mg_set_level_grid(self._h, self._idx(level), self.elements(level), n,
                                          ptr(gi) if gi is not None else None))

    def set_flat_level(self, A, prune_zeros: bool = True):
        """A single level holding any square CSR matrix (stand-alone smoother / residual)."""
        indptr, is64, indices, data = _csr_arrays(A)
        check(self._lib.mg_set_level_csr(self._h, 0, 0, A.shape[0], data.size, ptr(indptr), is64, ptr(indices),
                                         ptr(data), None, 1 if prune_zeros else 0))
        self._flat_n = A.shape[0]

    def time_kernel(self, kernel: str, level: int, reps: int = 20) -> float:
        ms = C.c_double()
        check(self._lib.mg_time_kernel(self._h, kernel.encode(), self._idx(level), int(reps), C.byref(ms)))
        return float(ms.value)

    def sync(self):
        check(self._lib.mg_sync(self._h))
