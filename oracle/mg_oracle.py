"""CPU oracle: a NumPy/SciPy restatement of the reference's V-cycle path.

TEST INFRASTRUCTURE ONLY.  Nothing under `oracle/` is part of the product: only
`tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may
import it, and there only as the checker / the reported CPU baseline.  The product
path (`multigrid_dolfinx_amd`) never imports this module and has no CPU fallback.

Parity status: PINNED for 2-D.  Every function here is checked bit-for-bit (or to
<= 1e-15) against golden vectors produced by importing the reference's unmodified
`multigrid.py` in the build container (`tests/golden/make_golden.py`, fixtures under
`tests/golden/*.npz`, test `tests/test_oracle_golden.py`).  The reference's own
tests pin nothing on this path.  3-D has no reference implementation
at all: the 3-D branches below are the dimension-consistent extension and are
"parity unpinned".

What is restated (reference file:line -> here):
  multigrid.py:48-56    getJacobiMatrices      -> get_jacobi_matrices
  multigrid.py:223-228  jacobiRelaxation       -> jacobi_relaxation
  multigrid.py:59-120   Interpolation2D        -> Oracle.interpolate
  multigrid.py:123-132  Restriction2D_direct   -> Oracle.restrict_direct
  multigrid.py:135-198  Restriction2D          -> Oracle.restrict_full_weighting
  multigrid.py:231-268  V_cycle_scheme         -> Oracle.v_cycle
  multigrid.py:271-307  FullMultiGrid          -> Oracle.full_multigrid (l2 / mass-norm stop test)
  multigrid.py:312-339  FullMultiGrid_test     -> Oracle.full_multigrid_test

The one structural difference: the reference links levels through coordinate
dictionaries (`Multigrid_prototype.py:68-74`); here every level carries
`grid_index[dof]` (lexicographic node index) and the transfers are index arithmetic
on the lexicographic grid, in the reference's order of floating-point operations.
Arithmetic is delegated to the same third-party routines the reference calls
(SciPy `csr_matvec`, CSR-DIA subtraction, SuperLU `spsolve`; unpinned versions in
the reference, NumPy 2.2.6 / SciPy 1.15.3 here).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.sparse.linalg import spsolve

__all__ = ["get_jacobi_matrices", "jacobi_relaxation", "rbgs_relaxation", "lattice9_colors", "Oracle"]


def get_jacobi_matrices(A_and_level):
    """`(A, level) -> (D^-1 (A - D), D^-1, level)`; follows `multigrid.py:48-56`."""
    A, level = A_and_level
    d = A.diagonal()
    off_diag = A - sp.diags(d, 0)
    d_inv = sp.diags(1 / d, 0)
    return (d_inv.dot(off_diag), d_inv, level)


def jacobi_relaxation(A_jac, v, f, nw, omega):
    """`nw` sweeps `v <- (1-w) v + w D^-1 f - w (D^-1 R) v`; follows `multigrid.py:223-228`.

    `D^-1 f` is recomputed in every sweep, as the reference does (App. A Q7); the
    argument `v` is not modified.
    """
    DinvR, Dinv = A_jac[0], A_jac[1]
    for _ in range(nw):
        v = (1 - omega) * v + omega * Dinv.dot(f) - omega * DinvR.dot(v)
    return v


def lattice9_colors(grid_index, N, dim):
    """Nine colours for P2 rows on the (N+1)^dim lattice (N even): the seven parity classes of the edge / face /
    body mid-points (colour = parity bits of (i, j, k), 1..7) and the vertices -- all coordinates even -- split red /
    black by (i/2 + j/2 + k/2) mod 2 into colours 0 and 8.  No two coupled unknowns of a pruned P2 (or P1) Poisson
    matrix share a colour (tests/test_oracle_golden.py checks that on the assembled matrices).  2-D lattices use the
    bits of (i, 0, k), matching the device's (nx, 1, nz) storage.  NO REFERENCE COUNTERPART."""
    n1 = N + 1
    g = np.asarray(grid_index, dtype=np.int64)
    i = g % n1
    if dim == 3:
        j, k = (g // n1) % n1, g // (n1 * n1)
    else:
        j, k = np.zeros_like(g), g // n1
    par = (i & 1) | ((j & 1) << 1) | ((k & 1) << 2)
    vertex = (((i >> 1) + (j >> 1) + (k >> 1)) & 1) * 8
    return np.where(par == 0, vertex, par)


def rbgs_relaxation(A, v, f, nw, omega, color):
    """`nw` multi-colour Gauss-Seidel (SOR factor `omega`) sweeps: per sweep the colours are visited in ascending
    order and the rows of one colour are relaxed in place, `v_i += omega (f_i - (A v)_i) / a_ii` (red-black: `color`
    in {0, 1}; P2: `lattice9_colors`).  NO REFERENCE COUNTERPART (BASELINE.json config 5 names the smoother; the
    reference only has Jacobi): parity unpinned."""
    v = np.array(v, dtype=np.float64, copy=True)
    d_inv = 1.0 / A.diagonal()
    rows = [np.flatnonzero(color == c) for c in np.unique(color)]
    parts = [A[r, :] for r in rows]
    for _ in range(nw):
        for r, Ar in zip(rows, parts):
            v[r, 0] = v[r, 0] + omega * d_inv[r] * (f[r, 0] - Ar.dot(v[:, 0]))
    return v


class Oracle:
    """The reference's module state (`multigrid.py:10-45`) plus per-level grid maps.

    `bag` is any object with the 16 `Var_initializer` fields
    (`Multigrid_prototype.py:15-32`); `grid_index[level][dof]` is the lexicographic
    node index of each DoF (derived from the DoF coordinates by the caller).
    """

    def __init__(self, bag, grid_index, dim=2):
        self.dim = dim
        self.c = bag.coarsest_level_elements_per_dim
        self.coarsest_level = bag.coarsest_level
        self.finest_level = bag.finest_level
        self.A_sp_dict = bag.A_sp_dict
        self.A_jacobi_sp_dict = bag.A_jacobi_sp_dict
        self.b_dict = bag.b_dict
        self.mu0, self.mu1, self.mu2, self.omega = bag.mu0, bag.mu1, bag.mu2, bag.omega
        self.grid_index = {l: np.asarray(g, dtype=np.int64) for l, g in grid_index.items()}
        self.dof_of_node = {}
        for l, g in self.grid_index.items():
            inv = np.empty_like(g)
            inv[g] = np.arange(g.size, dtype=np.int64)
            self.dof_of_node[l] = inv
        if not self.A_jacobi_sp_dict:
            for l, a in self.A_sp_dict.items():
                self.A_jacobi_sp_dict[l] = get_jacobi_matrices(a)
        self.residual_history = []

    # ---- grid helpers -------------------------------------------------------------
    def elements(self, level):
        return self.c * 2 ** level

    def n_dofs(self, level):
        return (self.elements(level) + 1) ** self.dim

    def _to_grid(self, vec, level):
        n1 = self.elements(level) + 1
        return np.asarray(vec).reshape(-1)[self.dof_of_node[level]].reshape((n1,) * self.dim)

    def _from_grid(self, grid, level):
        return grid.reshape(-1)[self.grid_index[level]].reshape(-1, 1)

    # ---- transfers ----------------------------------------------------------------
    def interpolate_table(self, vec_2h, level_coarse, table):
        """Prolongation from a `(count, offsets, weights)` table (`poisson.p2_prolongation_table`: the natural embedding of
        the coarse P2 space; the device's `prolong_table`): fine point (i, j, k) combines the coarse lattice points
        `2 * floor((i, j, k) / 4) + offsets[r][t]`, r = (i mod 4) + 4 (j mod 4) + 16 (k mod 4), summed in table order,
        multiply then add.  NO REFERENCE COUNTERPART."""
        count, offsets, weights = table
        lf = level_coarse + 1
        nc1, nf1 = self.elements(level_coarse) + 1, self.elements(lf) + 1
        gc = self._to_grid(vec_2h, level_coarse)                  # indexed [k][j][i] (3-D) or [k][i] (2-D)
        idx = np.arange(nf1)
        if self.dim == 3:
            K, J, I = np.meshgrid(idx, idx, idx, indexing="ij")
        else:
            K, I = np.meshgrid(idx, idx, indexing="ij")
            J = np.zeros_like(I)
        res = (I & 3) | ((J & 3) << 2 if self.dim == 3 else 0) | ((K & 3) << 4)
        bi, bj, bk = 2 * (I >> 2), (2 * (J >> 2) if self.dim == 3 else J), 2 * (K >> 2)
        out = np.zeros(I.shape)
        for r in np.unique(res):
            m = res == r
            acc = None
            for t in range(count[r]):
                a, b, c = bi[m] + offsets[r, t, 0], bj[m] + offsets[r, t, 1], bk[m] + offsets[r, t, 2]
                vals = gc[c, b, a] if self.dim == 3 else gc[c, a]
                term = weights[r, t] * vals
                acc = term if acc is None else acc + term
            out[m] = acc
        return self._from_grid(out, lf)

    def restrict_table(self, vec_h, level_fine, table):
        """Restriction from a `(count, offsets, weights)` table gathered per coarse point (`poisson.p2_restriction_table`:
        the transpose of the P2 prolongation; the device's `restrict_table`): interior coarse point A sums
        `weights[type][t] * r[2 A + offsets[type][t]]` over the interior fine points, in table order, multiply then add;
        boundary coarse points take the coincident fine value.  NO REFERENCE COUNTERPART."""
        count, offsets, weights = table
        lc = level_fine - 1
        nc1, nf1 = self.elements(lc) + 1, self.elements(level_fine) + 1
        gf = self._to_grid(vec_h, level_fine)
        idx = np.arange(nc1)
        if self.dim == 3:
            K, J, I = np.meshgrid(idx, idx, idx, indexing="ij")
        else:
            K, I = np.meshgrid(idx, idx, indexing="ij")
            J = np.zeros_like(I)
        bnd = (I == 0) | (I == nc1 - 1) | (K == 0) | (K == nc1 - 1)
        if self.dim == 3:
            bnd |= (J == 0) | (J == nc1 - 1)
        typ = (I & 1) | ((J & 1) << 1 if self.dim == 3 else 0) | ((K & 1) << 2)
        out = np.zeros(I.shape)
        out[bnd] = (gf[2 * K[bnd], 2 * J[bnd], 2 * I[bnd]] if self.dim == 3 else gf[2 * K[bnd], 2 * I[bnd]])
        for t in np.unique(typ[~bnd]):
            m = (typ == t) & ~bnd
            acc = np.zeros(int(m.sum()))
            started = np.zeros(int(m.sum()), dtype=bool)
            for e in range(count[t]):
                ii, jj, kk = 2 * I[m] + offsets[t, e, 0], 2 * J[m] + offsets[t, e, 1], 2 * K[m] + offsets[t, e, 2]
                ok = (ii > 0) & (ii < nf1 - 1) & (kk > 0) & (kk < nf1 - 1)
                if self.dim == 3:
                    ok &= (jj > 0) & (jj < nf1 - 1)
                ii, jj, kk = np.clip(ii, 0, nf1 - 1), np.clip(jj, 0, nf1 - 1), np.clip(kk, 0, nf1 - 1)
                vals = gf[kk, jj, ii] if self.dim == 3 else gf[kk, ii]
                term = weights[t, e] * vals
                acc = np.where(ok, np.where(started, acc + term, term), acc)
                started |= ok
            out[m] = acc
        return self._from_grid(out, lc)

    def interpolate(self, vec_2h, level_coarse):
        """Q1 prolongation coarse -> `level_coarse + 1`; follows `multigrid.py:59-120`.

        Coincident nodes copy (`:72-75`); nodes on a coarse edge take `0.5*(a+b)`
        (`:83-102`); cell centres `0.25*(c1+c2+c3+c4)` summed in the reference's order
        (x-neighbour before y-neighbour, `:109-118`).  3-D: `0.125*` the 8 corners,
        x fastest (no reference).
        """
        C = self._to_grid(vec_2h, level_coarse)
        nf1 = self.elements(level_coarse + 1) + 1
        F = np.zeros((nf1,) * self.dim)
        if self.dim == 2:
            F[::2, ::2] = C
            F[::2, 1::2] = 0.5 * (C[:, :-1] + C[:, 1:])
            F[1::2, ::2] = 0.5 * (C[:-1, :] + C[1:, :])
            F[1::2, 1::2] = 0.25 * (C[:-1, :-1] + C[:-1, 1:] + C[1:, :-1] + C[1:, 1:])
        else:
            lo, hi = slice(None, -1), slice(1, None)
            al = slice(None)
            for pk in (0, 1):
                for pj in (0, 1):
                    for pi in (0, 1):
                        terms = []
                        for dk in ((0, 1) if pk else (0,)):
                            for dj in ((0, 1) if pj else (0,)):
                                for di in ((0, 1) if pi else (0,)):
                                    sk = (hi if dk else lo) if pk else al
                                    sj = (hi if dj else lo) if pj else al
                                    si = (hi if di else lo) if pi else al
                                    terms.append(C[sk, sj, si])
                        acc = terms[0]
                        for t in terms[1:]:
                            acc = acc + t
                        if len(terms) > 1:
                            acc = (1.0 / len(terms)) * acc
                        F[pk::2, pj::2, pi::2] = acc
        return self._from_grid(F, level_coarse + 1)

    def restrict_direct(self, vec_h, level_fine):
        """Injection fine -> `level_fine - 1`; follows `multigrid.py:123-132`."""
        F = self._to_grid(vec_h, level_fine)
        C = F[(slice(None, None, 2),) * self.dim]
        return self._from_grid(np.ascontiguousarray(C), level_fine - 1)

    def restrict_full_weighting(self, vec_h, level_fine):
        """FD full weighting; follows `multigrid.py:135-198`.

        `(1/16) * (corners + 2*edges + 4*centre)`; neighbours outside the grid are
        skipped (`:172-194`), which equals adding 0.0 in the same order.  3-D (no
        reference): `(1/64) * (corners + 2*edges + 4*faces + 8*centre)`.
        """
        F = self._to_grid(vec_h, level_fine)
        nc1 = self.elements(level_fine - 1) + 1
        P = np.pad(F, 1)

        def at(*offs):      # fine value at (2I+di, 2J+dj[, 2K+dk]) for every coarse node
            sl = tuple(slice(1 + o, 1 + o + 2 * nc1 - 1, 2) for o in offs)
            return P[sl]

        if self.dim == 2:   # array axes are (j, i); the reference's tuples are (x=i, y=j)
            corners = 0 + at(-1, -1) + at(1, -1) + at(-1, 1) + at(1, 1)
            edges = 0 + at(-1, 0) + at(1, 0) + at(0, -1) + at(0, 1)
            C = (1 / 16) * (corners + 2 * edges + 4 * at(0, 0))
        else:
            sums = {1: 0, 2: 0, 3: 0}
            for dk in (-1, 0, 1):
                for dj in (-1, 0, 1):
                    for di in (-1, 0, 1):
                        m = abs(dk) + abs(dj) + abs(di)
                        if m:
                            sums[m] = sums[m] + at(dk, dj, di)
            C = (1 / 64) * (sums[3] + 2 * sums[2] + 4 * sums[1] + 8 * at(0, 0, 0))
        return self._from_grid(np.ascontiguousarray(C), level_fine - 1)

    # ---- cycles -------------------------------------------------------------------
    def coarse_solve(self, f_h):
        """Exact coarsest solve with a fresh factorisation; follows `multigrid.py:239-241`."""
        u = spsolve(self.A_sp_dict[self.coarsest_level][0], f_h)
        return np.array(u).reshape(len(u), 1)

    def smooth(self, A_h, v, f, nw, smoother):
        if smoother == "jacobi":
            return jacobi_relaxation(A_h, v, f, nw, self.omega)
        level = A_h[2]
        if smoother == "mcgs":
            color = lattice9_colors(self.grid_index[level], self.elements(level), self.dim)
        else:
            color = self.grid_index[level] & 1
        return rbgs_relaxation(self.A_sp_dict[level][0], v, f, nw, self.omega, color)

    def v_cycle(self, A_h, v_h, f_h, test=False, restriction="direct", smoother="jacobi"):
        """One recursive V(mu1, mu2) cycle; follows `multigrid.py:231-268`.

        The smoother uses the argument `A_h`, the residual and the coarsest solve use
        `A_sp_dict` (App. A Q3); the coarse initial guess is zero (`:253`); with `test`
        the finest level returns `(v_h, f_2h, v_2h, err_h)` (`:262-266`).
        """
        level = A_h[2]
        if level == self.coarsest_level:
            return self.coarse_solve(f_h)
        v_h = self.smooth(A_h, v_h, f_h, self.mu1, smoother)
        r_h = f_h - self.A_sp_dict[level][0].dot(v_h)
        if restriction == "direct":
            f_2h = self.restrict_direct(r_h, level)
        elif restriction == "table":
            f_2h = self.restrict_table(r_h, level, self.restriction_table)
        else:
            f_2h = self.restrict_full_weighting(r_h, level)
        v_2h = np.zeros((f_2h.shape[0], 1))
        v_2h = self.v_cycle(self.A_jacobi_sp_dict[level - 1], v_2h, f_2h, test, restriction, smoother)
        table = getattr(self, "prolongation_table", None)
        err_h = self.interpolate(v_2h, level - 1) if table is None else self.interpolate_table(v_2h, level - 1, table)
        v_h = v_h + err_h
        v_h = self.smooth(A_h, v_h, f_h, self.mu2, smoother)
        if test and level == self.finest_level:
            return v_h, f_2h, v_2h, err_h
        return v_h

    def full_multigrid_test(self, A_h, f_h, test=False):
        """FMG with exactly `mu0` cycles on every level; follows `multigrid.py:312-339`."""
        level = A_h[2]
        if level == self.coarsest_level:
            return self.coarse_solve(f_h)
        f_2h = self.b_dict[level - 1]
        v_2h = self.full_multigrid_test(self.A_jacobi_sp_dict[level - 1], f_2h, test)
        v_h = self.interpolate(v_2h, level - 1)
        extras = (None, None, None)
        for _ in range(self.mu0):
            if level == self.finest_level:
                out = self.v_cycle(A_h, v_h, f_h, test)
                if not test:
                    # Q9: the reference unpacks four values from the single (n, 1) array
                    # (`multigrid.py:331-333`), which raises for every n != 4.
                    raise ValueError("too many values to unpack (expected 4)")
                v_h, extras = out[0], out[1:]
            else:
                v_h = self.v_cycle(A_h, v_h, f_h)
        if test and level == self.finest_level:
            return (v_h,) + tuple(extras)
        return v_h

    def full_multigrid(self, A_h, f_h, tol=1e-11, max_cycles=10000, norm=None):
        """FMG driver; follows `multigrid.py:271-307`.

        The reference's stop test uses dolfinx-assembled L2(Omega) norms (`:292-296`),
        which cannot be evaluated without dolfinx; `norm(r)` defaults to the l2 norm
        and may be given as `sqrt(r^T M r)` with a caller-supplied mass matrix.  Per
        finest cycle the norm is appended to `residual_history` (`:294-295`).
        """
        if norm is None:
            norm = lambda r: float(np.sqrt(np.sum(r * r)))
        level = A_h[2]
        if level == self.coarsest_level:
            return self.coarse_solve(f_h)
        f_2h = self.b_dict[level - 1]
        v_2h = self.full_multigrid(self.A_jacobi_sp_dict[level - 1], f_2h, tol, max_cycles, norm)
        v_h = self.interpolate(v_2h, level - 1)
        if level == self.finest_level:
            for _ in range(max_cycles):
                v_h = self.v_cycle(A_h, v_h, f_h)
                res_h = f_h - self.A_sp_dict[level][0].dot(v_h)
                rn = norm(res_h)
                self.residual_history.append(rn)
                if rn <= tol:
                    break
            return v_h
        for _ in range(self.mu0):
            v_h = self.v_cycle(A_h, v_h, f_h)
        return v_h
