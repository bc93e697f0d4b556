"""Synthetic P1 Poisson hierarchies in the conventions of a dolfinx hand-off.

The reference builds its level inputs with dolfinx (`Multigrid_prototype.py:62-118`):
per level a `UnitSquareMesh`, a CG1 space, Dirichlet data `1 + x^2 + 2y^2` on the
whole boundary, `assemble_matrix(a, bcs=[bc])` exported with `getValuesCSR()` into a
SciPy CSR (`:92-96`), a lifted right-hand side reshaped to `(n, 1)` (`:100-110`) and a
two-way DoF<->coordinate dictionary (`:68-74`).  dolfinx is not installable here, so
this module synthesises inputs of exactly that shape:

* fp64 values, int32 `indices`/`indptr`, columns sorted within a row (PETSc AIJ),
* the full P1 sparsity pattern with the numerically-zero couplings kept as explicit
  0.0 (7 entries per interior row in 2-D, 15 in 3-D),
* symmetric Dirichlet treatment: boundary rows *and* columns zeroed, 1.0 on the
  boundary diagonal,
* right-hand side `f*h^d` on interior nodes, lifted by the boundary data, boundary
  entries set to the boundary data,
* an arbitrary DoF numbering (dolfinx numbering is not lexicographic and differs per
  level, `test/test_mesh.py:36`), related to the grid only through coordinates.

3-D has no reference counterpart; it is the dimension-consistent extension
(`g = 1 + x^2 + 2y^2 + 3z^2`, `f = -12`, Kuhn 6-tetrahedra split) of SURVEY.md §8(d).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import scipy.sparse as sp


def _offsets(dim: int):
    """P1 sparsity offsets (di, dj[, dk]) in ascending lexicographic-column order."""
    if dim == 2:
        offs = [(0, 0), (1, 0), (-1, 0), (0, 1), (0, -1), (1, 1), (-1, -1)]
        key = lambda o: (o[1], o[0])
    elif dim == 3:
        base = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (1, 1, 1)]
        offs = [(0, 0, 0)] + base + [tuple(-c for c in o) for o in base]
        key = lambda o: (o[2], o[1], o[0])
    else:
        raise ValueError("dim must be 2 or 3")
    return sorted(offs, key=key)


def boundary_data(coords: np.ndarray, dim: int) -> np.ndarray:
    """Dirichlet data / exact solution (`Multigrid_prototype.py:78`; 3-D extension)."""
    g = 1.0 + coords[:, 0] ** 2 + 2.0 * coords[:, 1] ** 2
    if dim == 3:
        g = g + 3.0 * coords[:, 2] ** 2
    return g


def source_term(dim: int) -> float:
    """Constant source `f` (`Multigrid_prototype.py:90`): -6 in 2-D, -12 in 3-D."""
    return -6.0 if dim == 2 else -12.0


@dataclass
class Level:
    """One level of a hierarchy, as the reference's script would hand it over."""
    N: int                      # elements per dimension
    dim: int
    A: sp.csr_matrix            # (n, n) fp64 / int32, explicit zeros kept, DoF numbering
    b: np.ndarray               # (n, 1) lifted right-hand side
    coords: np.ndarray          # (n, 3) DoF coordinates (z = 0 in 2-D), DoF numbering
    grid_index: np.ndarray      # (n,) int64: lexicographic node index of each DoF
    h: float

    @property
    def n(self) -> int:
        return self.A.shape[0]

    def exact(self) -> np.ndarray:
        return boundary_data(self.coords, self.dim).reshape(-1, 1)


def lexicographic_level(N: int, dim: int, keep_zeros: bool = True) -> Level:
    """Assemble one level in lexicographic numbering (x fastest)."""
    n1 = N + 1
    n = n1 ** dim
    h = 1.0 / N
    if n * 15 >= 2 ** 31 and dim == 3 and keep_zeros:
        raise ValueError("nnz would overflow int32 indptr; use the device generator")
    idx = np.arange(n, dtype=np.int64)
    ijk = [idx % n1, (idx // n1) % n1]
    if dim == 3:
        ijk.append(idx // (n1 * n1))
    on_bnd = np.zeros(n, dtype=bool)
    for c in ijk:
        on_bnd |= (c == 0) | (c == N)
    coords = np.zeros((n, 3))
    for d in range(dim):
        coords[:, d] = ijk[d] / N
    g = boundary_data(coords, dim)
    w = 1.0 if dim == 2 else h              # magnitude of an axis coupling
    diag = 4.0 if dim == 2 else 6.0 * h
    strides = [1, n1, n1 * n1][:dim]

    offs = _offsets(dim)
    valid, colv, valv = [], [], []
    b = np.where(on_bnd, g, source_term(dim) * h ** dim)
    for o in offs:
        ok = np.ones(n, dtype=bool)
        delta = 0
        for d in range(dim):
            if o[d] > 0:
                ok &= ijk[d] + o[d] <= N
            elif o[d] < 0:
                ok &= ijk[d] + o[d] >= 0
            delta += o[d] * strides[d]
        col = idx + delta
        nz_axis = sum(1 for c in o if c != 0)
        if nz_axis == 0:
            val = np.where(on_bnd, 1.0, diag)
        elif nz_axis == 1:
            colc = np.clip(col, 0, n - 1)
            both_int = ok & ~on_bnd & ~on_bnd[colc]
            val = np.where(both_int, -w, 0.0)
            lift = ok & ~on_bnd & on_bnd[colc]      # apply_lifting: b -= A_full[:, bc] g
            b = b + np.where(lift, w * g[colc], 0.0)
        else:
            val = np.zeros(n)
        if not keep_zeros:
            ok = ok & (val != 0.0)
        valid.append(ok)
        colv.append(col)
        valv.append(val)
    counts = np.zeros(n, dtype=np.int64)
    for ok in valid:
        counts += ok
    indptr = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    nnz = int(indptr[-1])
    indices = np.empty(nnz, dtype=np.int32)
    data = np.empty(nnz, dtype=np.float64)
    pos = indptr[:-1].copy()
    for ok, col, val in zip(valid, colv, valv):
        p = pos[ok]
        indices[p] = col[ok]
        data[p] = val[ok]
        pos += ok
    A = sp.csr_matrix((data, indices, indptr.astype(np.int32)), shape=(n, n))
    A.has_sorted_indices = True
    return Level(N=N, dim=dim, A=A, b=b.reshape(n, 1), coords=coords, grid_index=idx.copy(), h=h)


def renumber(level: Level, numbering: np.ndarray) -> Level:
    """Re-express a lexicographic level in the DoF numbering `numbering[p] = dof`."""
    n = level.n
    numbering = np.asarray(numbering, dtype=np.int64)
    grid_index = np.empty(n, dtype=np.int64)
    grid_index[numbering] = np.arange(n, dtype=np.int64)
    A = level.A
    rows = np.repeat(numbering, np.diff(A.indptr))
    cols = numbering[A.indices]
    order = np.lexsort((cols, rows))
    counts = np.bincount(rows, minlength=n)
    indptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(counts, out=indptr[1:])
    A2 = sp.csr_matrix((A.data[order], cols[order].astype(np.int32), indptr), shape=(n, n))
    A2.has_sorted_indices = True
    b = np.empty_like(level.b)
    b[numbering] = level.b
    coords = np.empty_like(level.coords)
    coords[numbering] = level.coords
    return Level(N=level.N, dim=level.dim, A=A2, b=b, coords=coords, grid_index=grid_index, h=level.h)


def make_level(N: int, dim: int, seed: Optional[int] = None, keep_zeros: bool = True) -> Level:
    """One level; `seed` selects a reproducible random DoF numbering (None = lexicographic)."""
    lvl = lexicographic_level(N, dim, keep_zeros=keep_zeros)
    if seed is None:
        return lvl
    rng = np.random.default_rng(seed + 7919 * N)
    return renumber(lvl, rng.permutation(lvl.n))


def mesh_dof_dict(level: Level, decimals: int = 9) -> dict:
    """The reference's two-way dictionary `{dof: xyz} U {xyz: dof}` (`Multigrid_prototype.py:69-74`)."""
    d = {}
    for j in range(level.n):
        tup = tuple(round(float(c), decimals) for c in level.coords[j])
        d[j] = tup
        d[tup] = j
    return d


def grid_index_from_coords(coords: np.ndarray, N: int, dim: int) -> np.ndarray:
    """Lexicographic node index of every DoF from its coordinates.

    This replaces the reference's coordinate hashing (`multigrid.py:71-72`, `:78-79`,
    `:129-130`); rounding to the nearest grid line is valid for any N, unlike the
    reference's `int(x / h)` truncation.
    """
    coords = np.asarray(coords, dtype=np.float64)
    n1 = N + 1
    ijk = np.rint(coords[:, :dim] * N).astype(np.int64)
    if ijk.min() < 0 or ijk.max() > N:
        raise ValueError("coordinates outside the unit square/cube")
    gi = ijk[:, 0] + n1 * ijk[:, 1]
    if dim == 3:
        gi = gi + n1 * n1 * ijk[:, 2]
    if np.unique(gi).size != gi.size:
        raise ValueError("coordinates do not identify distinct grid nodes")
    return gi


@dataclass
class Hierarchy:
    """The 16-field attribute bag of `Var_initializer` (`Multigrid_prototype.py:15-32`)."""
    mesh_dof_list_dict: Dict[int, dict]
    element_size: Dict[int, float]
    coarsest_level_elements_per_dim: int
    coarsest_level: int
    finest_level: int
    A_sp_dict: Dict[int, tuple]
    A_jacobi_sp_dict: Dict[int, tuple]
    b_dict: Dict[int, np.ndarray]
    mu0: int
    mu1: int
    mu2: int
    omega: float
    residual_per_V_cycle_finest: list = field(default_factory=list)
    error_per_V_cycle_finest: list = field(default_factory=list)
    u_exact_fine: object = None
    V_fine_dolfx: object = None
    # extras (not part of the reference bag): the generated levels and the dimension
    levels: Dict[int, Level] = field(default_factory=dict)
    dim: int = 2


def make_hierarchy(dim: int, coarsest_level: int, finest_level: int, c: int = 8,
                   mu0: int = 2, mu1: int = 50, mu2: int = 50, omega: float = 2.0 / 3.0,
                   seed: Optional[int] = None, with_dicts: bool = False,
                   keep_zeros: bool = True) -> Hierarchy:
    """Level loop of `Multigrid_prototype.py:62-118` on synthetic inputs (N_l = c * 2**l).

    `A_jacobi_sp_dict` is left empty: it is filled by whichever `getJacobiMatrices`
    (the reference's, the oracle's or the HIP module's) the caller is exercising,
    as the script does at `Multigrid_prototype.py:135-136`.
    """
    levels, dicts, hs, As, bs = {}, {}, {}, {}, {}
    for l in range(coarsest_level, finest_level + 1):
        N = c * 2 ** l
        lvl = make_level(N, dim, seed=seed, keep_zeros=keep_zeros)
        levels[l] = lvl
        hs[l] = 1.0 / N
        As[l] = (lvl.A, l)
        bs[l] = lvl.b
        if with_dicts:
            dicts[l] = mesh_dof_dict(lvl)
    return Hierarchy(mesh_dof_list_dict=dicts, element_size=hs, coarsest_level_elements_per_dim=c,
                     coarsest_level=coarsest_level, finest_level=finest_level, A_sp_dict=As,
                     A_jacobi_sp_dict={}, b_dict=bs, mu0=mu0, mu1=mu1, mu2=mu2, omega=omega,
                     levels=levels, dim=dim)


# ---- P2 elements (BASELINE.json config 5; no reference implementation exists) ---------------------------------
def _p2_local_stiffness(verts: np.ndarray) -> np.ndarray:
    """Stiffness of one P2 simplex (triangle: 6 nodes, tetrahedron: 10 nodes) with vertex coordinates
    `verts` ((d+1, d)); node order: vertices, then edge midpoints (a, b) for a < b."""
    d = verts.shape[1]
    nv = d + 1
    T = np.hstack([np.ones((nv, 1)), verts])
    grad_lam = np.linalg.inv(T)[1:, :].T                 # (nv, d): gradient of each barycentric coordinate
    vol = abs(np.linalg.det(T)) / (2.0 if d == 2 else 6.0)
    edges = [(a, b) for a in range(nv) for b in range(a + 1, nv)]
    # degree-2 quadrature: triangle -> edge midpoints (weights 1/3); tetrahedron -> 4-point rule
    if d == 2:
        pts = np.array([[0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5]])
        wts = np.full(3, 1.0 / 3.0)
    else:
        a, b = 0.5854101966249685, 0.1381966011250105
        pts = np.full((4, 4), b) + (a - b) * np.eye(4)
        wts = np.full(4, 0.25)
    nn = nv + len(edges)
    K = np.zeros((nn, nn))
    for lam, w in zip(pts, wts):
        G = np.zeros((nn, d))
        for i in range(nv):
            G[i] = (4.0 * lam[i] - 1.0) * grad_lam[i]
        for e, (i, j) in enumerate(edges):
            G[nv + e] = 4.0 * (lam[i] * grad_lam[j] + lam[j] * grad_lam[i])
        K += w * vol * (G @ G.T)
    return K


def p2_level(N: int, dim: int, seed: Optional[int] = None) -> Level:
    """P2 Poisson level on the structured simplicial mesh with N cells per dimension: the DoFs are ALL points
    of the (2N+1)^dim lattice (vertices, edge / face-diagonal / body-diagonal midpoints), so `Level.N` is 2N and
    the lattice nests under refinement like the P1 grids do.  Same problem, boundary treatment and hand-off
    conventions as `lexicographic_level`; the stencil reaches two lattice planes.  Because P2 reproduces the
    quadratic manufactured solution exactly, the discrete solution equals the nodal exact values."""
    M = 2 * N
    n1 = M + 1
    n = n1 ** dim
    h = 1.0 / N
    idx = np.arange(n, dtype=np.int64)
    ijk = [idx % n1, (idx // n1) % n1] + ([idx // (n1 * n1)] if dim == 3 else [])
    coords = np.zeros((n, 3))
    for d in range(dim):
        coords[:, d] = ijk[d] / M
    on_bnd = np.zeros(n, dtype=bool)
    for cdir in ijk:
        on_bnd |= (cdir == 0) | (cdir == M)
    g = boundary_data(coords, dim)
    # reference simplices of one cell (right diagonal in 2-D, Kuhn split in 3-D), vertex offsets in cells
    import itertools
    simplices = []
    for perm in itertools.permutations(range(dim)):
        v = [np.zeros(dim, dtype=int)]
        for ax in perm:
            nxt = v[-1].copy()
            nxt[ax] += 1
            v.append(nxt)
        simplices.append(np.array(v))
    strides = np.array([1, n1, n1 * n1][:dim])
    cells = np.stack(np.meshgrid(*[np.arange(N)] * dim, indexing="ij"), axis=-1).reshape(-1, dim)
    rows, cols, vals = [], [], []
    load = np.zeros(n)
    f = source_term(dim)
    for sv in simplices:
        K = _p2_local_stiffness(sv * h)
        nv = dim + 1
        edges = [(a, b) for a in range(nv) for b in range(a + 1, nv)]
        local = [2 * sv[a] for a in range(nv)] + [sv[a] + sv[b] for a, b in edges]       # lattice offsets
        node = np.stack([(2 * cells + off) @ strides for off in local], axis=1)           # (ncells, nn)
        nn = node.shape[1]
        rows.append(np.repeat(node, nn, axis=1).reshape(-1))
        cols.append(np.tile(node, (1, nn)).reshape(-1))
        vals.append(np.tile(K.reshape(-1), cells.shape[0]))
        # load vector of a constant source: vertices get 0 (2-D) / -vol/20 (3-D), edge nodes vol/3 / vol/5
        vol = h ** dim / (2.0 if dim == 2 else 6.0)
        wv, we = (0.0, vol / 3.0) if dim == 2 else (-vol / 20.0, vol / 5.0)
        for a in range(nn):
            np.add.at(load, node[:, a], f * (wv if a < nv else we))
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    # symmetric Dirichlet treatment with explicit zeros kept, as assemble_matrix(bcs=[bc]) does
    b = load - A[:, on_bnd] @ g[on_bnd]
    b[on_bnd] = g[on_bnd]
    A = A.tolil()
    bidx = np.flatnonzero(on_bnd)
    keep = sp.diags((~on_bnd).astype(float))
    A = (keep @ A.tocsr() @ keep + sp.diags(on_bnd.astype(float))).tocsr()
    A.sort_indices()
    A = sp.csr_matrix((A.data, A.indices.astype(np.int32), A.indptr.astype(np.int32)), shape=(n, n))
    lvl = Level(N=M, dim=dim, A=A, b=b.reshape(n, 1), coords=coords, grid_index=idx.copy(), h=h)
    if seed is None:
        return lvl
    rng = np.random.default_rng(seed + 7919 * N)
    return renumber(lvl, rng.permutation(n))


def p2_stencils(dim: int):
    """Interior rows of the P2 Poisson matrix per parity class of the lattice point, for mesh width h = 1 (cell
    size; lattice spacing 1/2): `(count[8], offsets[8, 64, 3], values[8, 64], load[8])` as `mg_gen_lattice_level` takes
    them, read off a small assembled level (`p2_level(4, dim)`, whose centre rows touch no boundary).  Values scale
    with h^(dim-2), the load with h^dim; for h a power of two both scalings are exact, so a level generated from these
    tables equals the assembled one bit for bit in its matrix."""
    N = 4
    L = p2_level(N, dim)
    M, n1 = L.N, L.N + 1
    h = 1.0 / N
    A = L.A.tocsr()
    count = np.zeros(8, dtype=np.int32)
    offsets = np.zeros((8, 64, 3), dtype=np.int32)
    values = np.zeros((8, 64))
    load = np.zeros(8)
    for cls in range(8):
        bits = (cls & 1, (cls >> 1) & 1, (cls >> 2) & 1)
        if dim == 2 and bits[1]:
            continue                                   # 2-D lattices are stored as (nx, 1, nz): j is always 0
        ijk = [4 + b for b in bits]                    # a centre node of this class
        if dim == 2:
            node = ijk[0] + n1 * ijk[2]
        else:
            node = ijk[0] + n1 * (ijk[1] + n1 * ijk[2])
        lo, hi = A.indptr[node], A.indptr[node + 1]
        cols, vals = A.indices[lo:hi], A.data[lo:hi]
        keep = vals != 0.0
        cols, vals = cols[keep], vals[keep]
        order = np.argsort(cols)
        cols, vals = cols[order], vals[order]
        count[cls] = cols.size
        for t, (cc, vv) in enumerate(zip(cols, vals)):
            ci = cc % n1
            cj = (cc // n1) % n1 if dim == 3 else 0
            ck = cc // (n1 * n1) if dim == 3 else cc // n1
            offsets[cls, t] = (ci - ijk[0], cj - ijk[1] if dim == 3 else 0, ck - ijk[2])
            values[cls, t] = vv / h ** (dim - 2)
        load[cls] = L.b[node, 0] / h ** dim
    return count, offsets, values, load


def p2_prolongation_table(dim: int):
    """The natural prolongation between nested P2 spaces on the structured simplicial meshes (coarse cells of four
    fine lattice steps per dimension): a fine lattice point takes the value of the coarse P2 function there.  That value
    only depends on the point's position inside its coarse cell, `(i mod 4, j mod 4, k mod 4)`, so the operator is a table
    `(count[64], offsets[64, 10, 3], weights[64, 10])`: for residue `ri + 4 rj + 16 rk` the `count` coarse lattice
    points `2 * floor((i, j, k) / 4) + offsets` (coarse lattice units, 0..2 per axis) and their weights -- the P2 basis
    functions `lambda_a (2 lambda_a - 1)` / `4 lambda_a lambda_b` of the Kuhn simplex that contains the point
    (`x_p1 >= x_p2 >= x_p3` for the simplex of axis permutation p, as `p2_level` splits the cells).  Coincident points
    copy, every other point combines up to 10 (3-D) / 6 (2-D) coarse values with weights like 3/8, 3/4, -1/8.
    2-D lattices use the residues with `rj = 0` (device storage `(nx, 1, nz)`).  NO REFERENCE COUNTERPART
    (`Interpolation2D` is bilinear, multigrid.py:59-120): BASELINE config 5's "higher-bandwidth transfer ops"."""
    import itertools
    axes = (0, 2) if dim == 2 else (0, 1, 2)            # lattice axes in use: (i, k) in 2-D
    count = np.zeros(64, dtype=np.int32)
    offsets = np.zeros((64, 10, 3), dtype=np.int32)
    weights = np.zeros((64, 10))
    for res in range(64):
        r = (res & 3, (res >> 2) & 3, (res >> 4) & 3)
        if dim == 2 and r[1]:
            continue
        x = np.array([r[a] / 4.0 for a in axes])         # position in the coarse cell
        perm = sorted(range(dim), key=lambda d: (-x[d], d))
        verts = [np.zeros(dim, dtype=int)]
        for d in perm:
            nxt = verts[-1].copy()
            nxt[d] += 1
            verts.append(nxt)
        xs = [x[d] for d in perm]
        lam = [1.0 - xs[0]] + [xs[t] - xs[t + 1] for t in range(dim - 1)] + [xs[-1]]
        entries = {}
        for a in range(dim + 1):
            w = lam[a] * (2.0 * lam[a] - 1.0)
            if w != 0.0:
                entries[tuple(2 * verts[a])] = entries.get(tuple(2 * verts[a]), 0.0) + w
        for a, b in itertools.combinations(range(dim + 1), 2):
            w = 4.0 * lam[a] * lam[b]
            if w != 0.0:
                key = tuple(verts[a] + verts[b])
                entries[key] = entries.get(key, 0.0) + w
        keys = sorted(entries, key=lambda kk: tuple(reversed(kk)))           # ascending coarse lexicographic order
        count[res] = len(keys)
        for t, kk in enumerate(keys):
            full = [0, 0, 0]
            for d, a in enumerate(axes):
                full[a] = kk[d]
            offsets[res, t] = full
            weights[res, t] = entries[kk]
    return count, offsets, weights


def p2_restriction_table(dim: int):
    """Transpose of `p2_prolongation_table`, gathered per coarse lattice point: `(count[8], offsets[8, M, 3],
    weights[8, M])` -- a coarse point A of type `(a & 1) + 2 (b & 1) + 4 (c & 1)` sums `weights[t] * r[2 A + offsets[t]]`
    over the fine lattice points whose prolongation uses it (the canonical finite-element restriction of a residual:
    `<R r, v> = <r, P v>`).  Entries in ascending fine lexicographic order.  NO REFERENCE COUNTERPART (the reference
    restricts by injection, multigrid.py:123-132, :251-252)."""
    count, offsets, weights = p2_prolongation_table(dim)
    fwd = {}
    for r in range(64):
        for t in range(count[r]):
            fwd[(r, tuple(int(x) for x in offsets[r, t]))] = weights[r, t]
    axes_on = (True, dim == 3, True)
    per_type = {}
    rng = range(-4, 5)
    for typ in range(8):
        t = (typ & 1, (typ >> 1) & 1, (typ >> 2) & 1)
        if dim == 2 and t[1]:
            continue
        ent = []
        for dk in rng:
            for dj in (rng if dim == 3 else (0,)):
                for di in rng:
                    d = (di, dj, dk)
                    res, need, ok = 0, [], True
                    for ax in range(3):
                        if not axes_on[ax]:
                            need.append(0)
                            continue
                        p = 2 * t[ax] + d[ax]                       # fine position relative to 4 q
                        res |= (p % 4) << (2 * ax)
                        o = t[ax] - 2 * (p // 4)                    # coarse offset that names A in that fine point's entry
                        if o < 0 or o > 2:
                            ok = False
                        need.append(o)
                    if ok and (res, tuple(need)) in fwd:
                        ent.append((d, fwd[(res, tuple(need))]))
        per_type[typ] = ent
    M = max(len(v) for v in per_type.values())
    rcount = np.zeros(8, dtype=np.int32)
    roff = np.zeros((8, M, 3), dtype=np.int32)
    rw = np.zeros((8, M))
    for typ, ent in per_type.items():
        rcount[typ] = len(ent)
        for e, (d, w) in enumerate(ent):
            roff[typ, e] = d
            rw[typ, e] = w
    return rcount, roff, rw
