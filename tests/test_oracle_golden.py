"""The oracle (`oracle/mg_oracle.py`) against golden vectors captured from the
reference's own `multigrid.py` (`tests/golden/make_golden.py`).  CPU only.

Tolerances: the oracle calls the same SciPy/NumPy routines in the same order as the
reference, so every comparison below is exact (== 0) unless stated.
"""
import hashlib

import numpy as np
import pytest

from oracle.mg_oracle import Oracle, get_jacobi_matrices, jacobi_relaxation
from tests.helpers import bag_from_fixture, hierarchy_for, load_golden

FULL = ["c1_lex", "c1_perm"]


def _oracle(name):
    g = load_golden(name)
    bag, grid_index, coords = bag_from_fixture(g)
    return g, bag, Oracle(bag, grid_index, dim=2)


@pytest.mark.parametrize("name", FULL)
def test_generator_reproduces_fixture_inputs(name):
    """The committed inputs are what `poisson.make_hierarchy` generates today."""
    from multigrid_dolfinx_amd import poisson
    g = load_golden(name)
    h = poisson.make_hierarchy(2, 1, 3, seed=None if name.endswith("lex") else 0)
    for l in (1, 2, 3):
        A = h.levels[l].A
        assert np.array_equal(A.indptr, g[f"A{l}_indptr"]) and A.indptr.dtype == np.int32
        assert np.array_equal(A.indices, g[f"A{l}_indices"]) and A.indices.dtype == np.int32
        assert np.array_equal(A.data, g[f"A{l}_data"])
        assert np.array_equal(h.levels[l].b, g[f"b{l}"])
        assert np.array_equal(h.levels[l].coords, g[f"coords{l}"])
        assert np.array_equal(poisson.grid_index_from_coords(g[f"coords{l}"], 8 * 2 ** l, 2), g[f"grid_index{l}"])


@pytest.mark.parametrize("name", FULL)
def test_get_jacobi_matrices(name):
    g, bag, orc = _oracle(name)
    for l in (1, 2, 3):
        R, Dinv, lev = get_jacobi_matrices(bag.A_sp_dict[l])
        assert lev == l
        assert np.array_equal(R.indptr, g[f"J{l}_indptr"])
        assert np.array_equal(R.indices, g[f"J{l}_indices"])
        assert np.array_equal(R.data, g[f"J{l}_data"])
        assert np.array_equal(Dinv.diagonal(), g[f"Dinv{l}"])
        assert bag.A_sp_dict[l][0].nnz == g[f"A{l}_data"].size      # A untouched (Q7)


@pytest.mark.parametrize("name", FULL)
def test_jacobi_relaxation(name):
    g, bag, orc = _oracle(name)
    v0 = g["jac_v0"].copy()
    for nw in (1, 50):
        out = jacobi_relaxation(orc.A_jacobi_sp_dict[3], v0, bag.b_dict[3], nw, bag.omega)
        assert out.shape == (4225, 1)
        assert np.array_equal(out, g[f"jac_nw{nw}"])
    assert np.array_equal(v0, g["jac_v0"])                          # input not mutated


@pytest.mark.parametrize("name", FULL)
def test_transfers(name):
    g, bag, orc = _oracle(name)
    for lc in (1, 2):
        lf = lc + 1
        assert np.array_equal(orc.interpolate(g[f"xfer_xc{lc}"], lc), g[f"interp_{lc}to{lf}"])
        assert np.array_equal(orc.restrict_direct(g[f"xfer_xf{lf}"], lf), g[f"inject_{lf}to{lc}"])
        assert np.array_equal(orc.restrict_full_weighting(g[f"xfer_xf{lf}"], lf), g[f"fullw_{lf}to{lc}"])


@pytest.mark.parametrize("name", FULL)
def test_v_cycle(name):
    g, bag, orc = _oracle(name)
    A3, f = orc.A_jacobi_sp_dict[3], bag.b_dict[3]
    v = np.zeros_like(f)
    for k in (1, 2, 3):
        v = orc.v_cycle(A3, v, f)
        assert v.shape == (4225, 1)
        assert np.array_equal(v, g[f"vcycle_iter{k}"])
        r = f - bag.A_sp_dict[3][0].dot(v)
        assert abs(np.sqrt(np.sum(r * r)) - g["vcycle_res_l2"][k - 1]) <= 1e-15 * g["vcycle_res_l2"][k - 1]
    t = orc.v_cycle(A3, np.zeros_like(f), f, True)
    for key, arr in zip(("v_h", "f_2h", "v_2h", "err_h"), t):
        assert np.array_equal(arr, g[f"vcycle_test_{key}"]), key
    mid = orc.v_cycle(orc.A_jacobi_sp_dict[2], np.zeros_like(bag.b_dict[2]), bag.b_dict[2], True)
    assert np.array_equal(mid, g["vcycle_mid_level2"])              # test only affects the finest (Q4)


@pytest.mark.parametrize("name", FULL)
def test_full_multigrid_test(name):
    g, bag, orc = _oracle(name)
    t = orc.full_multigrid_test(orc.A_jacobi_sp_dict[3], bag.b_dict[3], True)
    shapes = [a.shape for a in t]
    assert shapes == [(4225, 1), (1089, 1), (1089, 1), (4225, 1)]   # Multigrid_prototype.py:144-147
    for key, arr in zip(("v_h", "f_2h", "v_2h", "err_h"), t):
        assert np.array_equal(arr, g[f"fmg_test_{key}"]), key
    assert str(g["fmg_test_false_raises"]).startswith("ValueError")
    with pytest.raises(ValueError):                                  # Q9
        orc.full_multigrid_test(orc.A_jacobi_sp_dict[3], bag.b_dict[3], False)


def _sha(h):
    m = hashlib.sha256()
    for l in sorted(h.levels):
        A = h.levels[l].A
        for arr in (A.indptr, A.indices, A.data, h.levels[l].b, h.levels[l].grid_index):
            m.update(np.ascontiguousarray(arr).tobytes())
    return m.hexdigest()


@pytest.mark.parametrize("name", ["n128_mu2_perm", "n256_mu50_lex", "n512_mu2_lex"])
def test_larger_hierarchies(name):
    g = load_golden(name)
    h = hierarchy_for(g)
    assert _sha(h) == str(g["inputs_sha256"])
    orc = Oracle(h, {l: L.grid_index for l, L in h.levels.items()}, dim=2)
    hi = h.finest_level
    stride = int(g["meta_stride"])
    f = h.b_dict[hi]
    v = np.zeros_like(f)
    for k in range(1, g["vcycle_res_l2"].size + 1):
        v = orc.v_cycle(orc.A_jacobi_sp_dict[hi], v, f)
        assert np.array_equal(v[::stride], g[f"vcycle_iter{k}"])
        r = f - h.A_sp_dict[hi][0].dot(v)
        assert abs(np.sqrt(np.sum(r * r)) - g["vcycle_res_l2"][k - 1]) <= 1e-14 * g["vcycle_res_l2"][k - 1]
    if "fmg_test_v_h" in g.files:
        t = orc.full_multigrid_test(orc.A_jacobi_sp_dict[hi], f, True)
        for key, arr in zip(("v_h", "f_2h", "v_2h", "err_h"), t):
            assert np.array_equal(arr[::stride], g[f"fmg_test_{key}"]), key


def test_manufactured_solution_is_discrete_solution():
    """u = 1 + x^2 + 2y^2 (+3z^2) solves the synthetic systems to round-off."""
    from scipy.sparse.linalg import spsolve
    from multigrid_dolfinx_amd import poisson
    for dim, N in ((2, 32), (3, 8)):
        L = poisson.make_level(N, dim, seed=3)
        u = spsolve(L.A.tocsc(), L.b.ravel())
        assert np.abs(u - L.exact().ravel()).max() < 1e-12


def test_interpolation_reproduces_multilinear_and_injection_is_exact():
    from multigrid_dolfinx_amd import poisson
    for dim in (2, 3):
        h = poisson.make_hierarchy(dim, 0, 1, c=4, seed=5)
        orc = Oracle(h, {l: L.grid_index for l, L in h.levels.items()}, dim=dim)
        def q(c):
            out = 1.0 + 2 * c[:, 0] - c[:, 1] + 3 * c[:, 0] * c[:, 1]
            if dim == 3:
                out = out + 0.5 * c[:, 2] * (1 + c[:, 0] * c[:, 1])
            return out.reshape(-1, 1)
        qc, qf = q(h.levels[0].coords), q(h.levels[1].coords)
        assert np.abs(orc.interpolate(qc, 0) - qf).max() < 1e-14
        assert np.array_equal(orc.restrict_direct(qf, 1), qc)
        ones = np.ones_like(qf)
        fw = orc.restrict_full_weighting(ones, 1)
        interior = np.ones(h.levels[0].n, bool)
        for d in range(dim):
            interior &= (h.levels[0].coords[:, d] > 0) & (h.levels[0].coords[:, d] < 1)
        assert np.allclose(fw[interior], 1.0)


def test_rbgs_oracle_against_plain_loops():
    """Red-black Gauss-Seidel has no reference counterpart; pin the vectorised oracle to a scalar loop."""
    from multigrid_dolfinx_amd import poisson
    from oracle.mg_oracle import rbgs_relaxation
    for dim, N in ((2, 6), (3, 4)):
        L = poisson.make_level(N, dim, seed=2, keep_zeros=False)
        A = L.A.toarray()
        color = L.grid_index & 1
        rng = np.random.default_rng(0)
        v0 = rng.standard_normal((L.n, 1))
        got = rbgs_relaxation(L.A, v0, L.b, 2, 1.15, color)
        v = v0.copy()
        for _ in range(2):
            for c in (0, 1):
                for i in np.flatnonzero(color == c):
                    assert all(color[j] != c for j in np.flatnonzero(A[i]) if j != i)    # bipartite
                    v[i, 0] += 1.15 * (L.b[i, 0] - A[i] @ v[:, 0]) / A[i, i]
        assert np.abs(got - v).max() <= 1e-13
        assert np.array_equal(v0, v0)


def test_nine_colour_gauss_seidel_oracle_against_plain_loops():
    """BASELINE config 5's smoother on P2 rows (no reference: parity unpinned).  The nine lattice colours must be a
    proper colouring of the pruned P2 matrices (2-D and 3-D, lexicographic and permuted numbering), and the
    vectorised oracle must equal a scalar Gauss-Seidel loop that visits the rows colour by colour."""
    from multigrid_dolfinx_amd import poisson
    from oracle.mg_oracle import lattice9_colors, rbgs_relaxation
    for dim, N, seed in ((2, 4, None), (2, 3, 1), (3, 2, None), (3, 3, 2)):
        L = poisson.p2_level(N, dim, seed=seed)
        A = L.A.copy()
        A.eliminate_zeros()
        color = lattice9_colors(L.grid_index, L.N, dim)
        assert set(np.unique(color)) <= set(range(9)) and len(np.unique(color)) == (5 if dim == 2 else 9)
        D = A.toarray()
        rng = np.random.default_rng(0)
        v0 = rng.standard_normal((L.n, 1))
        got = rbgs_relaxation(A, v0, L.b, 2, 1.1, color)
        v = v0.copy()
        for _ in range(2):
            for c in np.unique(color):
                for i in np.flatnonzero(color == c):
                    # a proper colouring (but for round-off couplings of 1e-18 a_ii where h is not a power of two)
                    assert all(color[j] != c for j in np.flatnonzero(np.abs(D[i]) > 1e-12 * D[i, i]) if j != i)
                    v[i, 0] += 1.1 * (L.b[i, 0] - D[i] @ v[:, 0]) / D[i, i]
        assert np.abs(got - v).max() <= 1e-12 * max(1.0, np.abs(v).max())
    # the same colours are proper for the pruned P1 seven-point rows as well
    for dim, N in ((2, 6), (3, 4)):
        L = poisson.make_level(N, dim, seed=2, keep_zeros=False)
        color = lattice9_colors(L.grid_index, N, dim)
        C = L.A.tocoo()
        off = C.row != C.col
        assert not np.any(color[C.row[off]] == color[C.col[off]])


def test_p2_prolongation_table_reproduces_quadratics():
    """The table prolongation for nested P2 spaces (no reference counterpart) is pinned by what it is: the coarse P2
    function evaluated at the fine lattice points.  P2 contains every quadratic, so nodal values of a random quadratic
    on the coarse lattice must prolong to its nodal values on the fine lattice; coincident points copy; the weights of
    every point sum to one."""
    import types
    from multigrid_dolfinx_amd import poisson
    from oracle.mg_oracle import Oracle
    for dim in (2, 3):
        count, offsets, weights = poisson.p2_prolongation_table(dim)
        used = [r for r in range(64) if count[r]]
        assert len(used) == (16 if dim == 2 else 64)
        assert all(abs(weights[r, :count[r]].sum() - 1.0) <= 1e-15 for r in used)
        assert count[0] == 1 and weights[0, 0] == 1.0 and tuple(offsets[0, 0]) == (0, 0, 0)
        assert count.max() == (5 if dim == 2 else 10) and weights.min() == -0.125
        rng = np.random.default_rng(dim)
        co = rng.standard_normal(10)

        def q(c):
            x, y, z = c[:, 0], c[:, 1], c[:, 2]
            return (co[0] + co[1] * x + co[2] * y + co[3] * z + co[4] * x * x + co[5] * y * y + co[6] * z * z
                    + co[7] * x * y + co[8] * x * z + co[9] * y * z).reshape(-1, 1)
        levels = {l: poisson.p2_level(N, dim, seed=3 + l) for l, N in enumerate((2, 4))}
        bag = types.SimpleNamespace(
            mesh_dof_list_dict={}, element_size={}, coarsest_level_elements_per_dim=4, coarsest_level=0, finest_level=1,
            A_sp_dict={l: (L.A, l) for l, L in levels.items()}, A_jacobi_sp_dict={}, b_dict={l: L.b for l, L in levels.items()},
            mu0=1, mu1=1, mu2=1, omega=1.0, residual_per_V_cycle_finest=[], error_per_V_cycle_finest=[], u_exact_fine=None,
            V_fine_dolfx=None)
        orc = Oracle(bag, {l: L.grid_index for l, L in levels.items()}, dim=dim)
        fine = orc.interpolate_table(q(levels[0].coords), 0, (count, offsets, weights))
        assert np.abs(fine - q(levels[1].coords)).max() <= 1e-14
        # the reference's bilinear table is exact for bilinear functions only
        co[4:7] = 0.0
        assert np.abs(orc.interpolate(q(levels[0].coords), 0) - q(levels[1].coords)).max() <= 1e-14


def test_p2_restriction_table_is_the_transpose_of_the_prolongation():
    """`<R r, v> = <r, P v>` for random vectors that vanish on the boundary (where the lifted system decouples and the
    restriction injects): pins the gathered restriction table, its boundary handling and its index arithmetic to the
    prolongation, which is itself pinned to quadratic reproduction."""
    import types
    from multigrid_dolfinx_amd import poisson
    from oracle.mg_oracle import Oracle
    for dim in (2, 3):
        ptab, rtab = poisson.p2_prolongation_table(dim), poisson.p2_restriction_table(dim)
        levels = {l: poisson.p2_level(N, dim, seed=5 + l) for l, N in enumerate((2, 4))}
        bag = types.SimpleNamespace(
            mesh_dof_list_dict={}, element_size={}, coarsest_level_elements_per_dim=4, coarsest_level=0, finest_level=1,
            A_sp_dict={l: (L.A, l) for l, L in levels.items()}, A_jacobi_sp_dict={}, b_dict={l: L.b for l, L in levels.items()},
            mu0=1, mu1=1, mu2=1, omega=1.0, residual_per_V_cycle_finest=[], error_per_V_cycle_finest=[], u_exact_fine=None,
            V_fine_dolfx=None)
        orc = Oracle(bag, {l: L.grid_index for l, L in levels.items()}, dim=dim)
        rng = np.random.default_rng(dim)

        def interior(L):
            c = L.coords[:, :dim]
            return np.all((c > 1e-12) & (c < 1 - 1e-12), axis=1).reshape(-1, 1)
        r = rng.standard_normal((levels[1].n, 1)) * interior(levels[1])
        v = rng.standard_normal((levels[0].n, 1)) * interior(levels[0])
        lhs = float((orc.restrict_table(r, 1, rtab) * v).sum())
        rhs = float((r * orc.interpolate_table(v, 0, ptab)).sum())
        assert abs(lhs - rhs) <= 1e-12 * max(1.0, abs(rhs))
        # boundary coarse points inject
        r2 = rng.standard_normal((levels[1].n, 1))
        got = orc.restrict_table(r2, 1, rtab)
        inj = orc.restrict_direct(r2, 1)
        b = ~interior(levels[0]).ravel()
        assert np.array_equal(got[b], inj[b])
