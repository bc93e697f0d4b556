"""GPU parity tests: the HIP path (through the C ABI) against
  (1) golden vectors captured from the reference's own multigrid.py (tests/golden/*.npz),
  (2) the CPU oracle on the same seeded inputs (3-D, which has no reference: parity unpinned),
  (3) size-independent properties at larger sizes.

Tolerances (BASELINE.json north_star: residual/iterate match <= 1e-10 relative l2):
  TOL_ITER = 1e-10   V-cycle / FMG iterates and residual norms
  TOL_SWEEP = 1e-12  Jacobi sweeps (one-matrix form differs from the split form by ~2e-16 per sweep)
  exact (array_equal) for getJacobiMatrices and the three transfer operators.

The FMG 4-tuple holds a residual and two corrections of an almost converged iterate
(||f_2h|| ~ 1e-4 next to ||A|| ||v_h|| ~ 5e2 in the C1 fixture): such a residual is a difference of
numbers seven orders of magnitude larger, so ANY evaluation order -- the reference's included -- only
defines it to ~eps * ||A|| ||v|| / ||r|| ~ 5e-10 of its own norm.  Those entries are therefore
compared at TOL_ITER relative to the quantity they perturb (||f_h|| for the restricted residual,
||v_h|| for the corrections) and at TOL_CANCEL = 1e-7 relative to their own norm.
"""
import numpy as np
import pytest

from multigrid_dolfinx_amd import poisson
from tests.helpers import bag_from_fixture, hierarchy_for, load_golden, rel_l2

pytestmark = pytest.mark.gpu

TOL_ITER = 1e-10
TOL_SWEEP = 1e-12
TOL_CANCEL = 1e-7
FULL = ["c1_lex", "c1_perm"]


@pytest.fixture()
def mg():
    """The drop-in module with clean state."""
    from multigrid_dolfinx_amd import multigrid as m
    clean = dict(dim=2, prune_zeros=True, restriction="direct", smoother="jacobi", grid_index=None, tuning={}, norm="auto",
                 stop_tol=1e-11, max_cycles=10000)
    m.configure(**clean)
    yield m
    m.configure(**clean)


def _init_from_fixture(m, name, with_dicts=True):
    g = load_golden(name)
    bag, grid_index, coords = bag_from_fixture(g)
    if with_dicts:
        for l, c in coords.items():
            lvl = poisson.Level(N=8 * 2 ** l, dim=2, A=bag.A_sp_dict[l][0], b=bag.b_dict[l], coords=c,
                                grid_index=grid_index[l], h=1.0 / (8 * 2 ** l))
            bag.mesh_dof_list_dict[l] = poisson.mesh_dof_dict(lvl)
    else:
        m.configure(grid_index=grid_index)
    for l, a in bag.A_sp_dict.items():
        bag.A_jacobi_sp_dict[l] = m.getJacobiMatrices(a)
    m.initialize_problem(bag)
    return g, bag


def _check_fmg_tuple(got, want, f_h):
    """(v_h, f_2h, v_2h, err_h): see the module docstring for the two scales."""
    v_scale, f_scale = np.linalg.norm(want[0]), np.linalg.norm(f_h)
    assert rel_l2(got[0], want[0]) <= TOL_ITER
    for a, b, scale, key in ((got[1], want[1], f_scale, "f_2h"), (got[2], want[2], v_scale, "v_2h"),
                             (got[3], want[3], v_scale, "err_h")):
        assert np.linalg.norm(np.ravel(a) - np.ravel(b)) <= TOL_ITER * scale, key
        assert rel_l2(a, b) <= TOL_CANCEL, key


# ---- golden vectors from the reference -------------------------------------------------------------------
@pytest.mark.parametrize("name", FULL)
def test_get_jacobi_matrices_matches_reference(mg, name):
    g = load_golden(name)
    bag, _, _ = bag_from_fixture(g)
    for l in (1, 2, 3):
        R, Dinv, lev = mg.getJacobiMatrices(bag.A_sp_dict[l])
        assert lev == l
        assert R.indices.dtype == np.int32 and not R.has_sorted_indices
        assert np.array_equal(R.indptr, g[f"J{l}_indptr"])
        assert np.array_equal(R.indices, g[f"J{l}_indices"])
        assert np.array_equal(R.data, g[f"J{l}_data"])
        assert np.array_equal(Dinv.diagonal(), g[f"Dinv{l}"])
        assert bag.A_sp_dict[l][0].nnz == g[f"A{l}_data"].size


@pytest.mark.parametrize("name", FULL)
def test_transfer_operators_match_reference_exactly(mg, name):
    g, bag = _init_from_fixture(mg, name)
    d = bag.mesh_dof_list_dict
    hs = bag.element_size
    for lc in (1, 2):
        lf = lc + 1
        out = mg.Interpolation2D(g[f"xfer_xc{lc}"], d[lc], d[lf], hs[lc], hs[lf], g[f"xfer_xf{lf}"].shape[0])
        assert out.shape == g[f"interp_{lc}to{lf}"].shape
        assert np.array_equal(out, g[f"interp_{lc}to{lf}"])
        out = mg.Restriction2D_direct(g[f"xfer_xf{lf}"], d[lc], d[lf], g[f"xfer_xc{lc}"].shape[0])
        assert np.array_equal(out, g[f"inject_{lf}to{lc}"])
        out = mg.Restriction2D(g[f"xfer_xf{lf}"], d[lc], d[lf], hs[lc], hs[lf], g[f"xfer_xc{lc}"].shape[0])
        assert np.array_equal(out, g[f"fullw_{lf}to{lc}"])


@pytest.mark.parametrize("name", FULL)
def test_jacobi_relaxation_matches_reference(mg, name):
    g, bag = _init_from_fixture(mg, name)
    v0 = g["jac_v0"].copy()
    for nw in (1, 50):
        out = mg.jacobiRelaxation(bag.A_jacobi_sp_dict[3], v0, bag.b_dict[3], nw)
        assert out.shape == (4225, 1)
        assert rel_l2(out, g[f"jac_nw{nw}"]) <= TOL_SWEEP
    assert np.array_equal(v0, g["jac_v0"])                  # input not mutated
    assert np.array_equal(mg.jacobiRelaxation(bag.A_jacobi_sp_dict[3], v0, bag.b_dict[3], 0), v0)


@pytest.mark.parametrize("name", FULL)
def test_jacobi_relaxation_standalone_operands(mg, name):
    """Operands that do not belong to an initialised problem (the reference's function is pure)."""
    g = load_golden(name)
    bag, _, _ = bag_from_fixture(g)
    A = mg.getJacobiMatrices(bag.A_sp_dict[3])
    mg.omega = float(g["meta_omega"])
    out = mg.jacobiRelaxation(A, g["jac_v0"], bag.b_dict[3], 50)
    assert rel_l2(out, g["jac_nw50"]) <= TOL_SWEEP


@pytest.mark.parametrize("name", FULL)
@pytest.mark.parametrize("with_dicts", [True, False])
def test_v_cycle_matches_reference(mg, name, with_dicts):
    g, bag = _init_from_fixture(mg, name, with_dicts)
    A3, f = bag.A_jacobi_sp_dict[3], bag.b_dict[3]
    v = np.zeros_like(f)
    for k in (1, 2, 3):
        v_in, v_before, f_before = v, v.copy(), f.copy()
        v = mg.V_cycle_scheme(A3, v_in, f)
        assert v.shape == (4225, 1) and v is not v_in
        assert np.array_equal(v_in, v_before) and np.array_equal(f, f_before)   # inputs not mutated (multigrid.py:226-227)
        assert rel_l2(v, g[f"vcycle_iter{k}"]) <= TOL_ITER
        r = f - bag.A_sp_dict[3][0].dot(v)
        assert abs(np.linalg.norm(r) - g["vcycle_res_l2"][k - 1]) <= TOL_ITER * g["vcycle_res_l2"][k - 1]
    t = mg.V_cycle_scheme(A3, np.zeros_like(f), f, True)
    assert [a.shape for a in t] == [(4225, 1), (1089, 1), (1089, 1), (4225, 1)]
    for key, arr in zip(("v_h", "f_2h", "v_2h", "err_h"), t):
        assert rel_l2(arr, g[f"vcycle_test_{key}"]) <= TOL_ITER, key
    mid = mg.V_cycle_scheme(bag.A_jacobi_sp_dict[2], np.zeros_like(bag.b_dict[2]), bag.b_dict[2], True)
    assert isinstance(mid, np.ndarray)                       # test only changes the finest level (Q4)
    assert rel_l2(mid, g["vcycle_mid_level2"]) <= TOL_ITER


@pytest.mark.parametrize("name", FULL)
def test_full_multigrid_test_matches_reference(mg, name):
    g, bag = _init_from_fixture(mg, name)
    t = mg.FullMultiGrid_test(bag.A_jacobi_sp_dict[3], bag.b_dict[3], True)
    assert [a.shape for a in t] == [(4225, 1), (1089, 1), (1089, 1), (4225, 1)]   # Multigrid_prototype.py:144-147
    _check_fmg_tuple(t, [g[f"fmg_test_{k}"] for k in ("v_h", "f_2h", "v_2h", "err_h")], bag.b_dict[3])
    with pytest.raises(ValueError):                          # reference quirk Q9
        mg.FullMultiGrid_test(bag.A_jacobi_sp_dict[3], bag.b_dict[3], False)


@pytest.mark.parametrize("name", ["n128_mu2_perm", "n256_mu50_lex", "n512_mu2_lex"])
def test_larger_hierarchies_match_reference(mg, name):
    g = load_golden(name)
    h = hierarchy_for(g)
    mg.configure(grid_index={l: L.grid_index for l, L in h.levels.items()})
    for l, a in h.A_sp_dict.items():
        h.A_jacobi_sp_dict[l] = (None, None, l)              # only the level key is used by the shim
    mg.initialize_problem(h)
    hi, stride = h.finest_level, int(g["meta_stride"])
    f = h.b_dict[hi]
    v = np.zeros_like(f)
    for k in range(1, g["vcycle_res_l2"].size + 1):
        v = mg.V_cycle_scheme(h.A_jacobi_sp_dict[hi], v, f)
        assert np.linalg.norm(v[::stride] - g[f"vcycle_iter{k}"]) <= TOL_ITER * g["vcycle_l2"][k - 1]
        r = f - h.A_sp_dict[hi][0].dot(v)
        assert abs(np.linalg.norm(r) - g["vcycle_res_l2"][k - 1]) <= TOL_ITER * g["vcycle_res_l2"][k - 1]
    if "fmg_test_v_h" in g.files:
        t = mg.FullMultiGrid_test(h.A_jacobi_sp_dict[hi], f, True)
        _check_fmg_tuple([a[::stride] for a in t], [g[f"fmg_test_{k}"] for k in ("v_h", "f_2h", "v_2h", "err_h")], f)


# ---- oracle comparisons (3-D has no reference: parity unpinned) -----------------------------------------------
def _oracle_and_device(dim, lo, hi, c, seed, mu, **kw):
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    from oracle.mg_oracle import Oracle
    bag = poisson.make_hierarchy(dim, lo, hi, c=c, mu0=2, mu1=mu, mu2=mu, seed=seed)
    gi = {l: L.grid_index for l, L in bag.levels.items()}
    orc = Oracle(bag, gi, dim=dim)
    dev = DeviceHierarchy.from_bag(bag, dim=dim, grid_index=gi, **kw)
    return bag, orc, dev


@pytest.mark.parametrize("seed", [None, 2])
@pytest.mark.parametrize("prune", [True, False])
def test_3d_v_cycle_and_fmg_match_oracle(seed, prune):
    bag, orc, dev = _oracle_and_device(3, 1, 3, 2, seed, 3, prune_zeros=prune)     # N = 4, 8, 16
    with dev:
        f = bag.b_dict[3]
        v = np.zeros_like(f)
        dev.set_vector(3, "v", v)
        dev.set_vector(3, "f", f)
        for k in range(2):
            v = orc.v_cycle(orc.A_jacobi_sp_dict[3], v, f)
            res = dev.vcycle(3, 1, residuals=True)
            assert rel_l2(dev.get_vector(3, "v"), v) <= TOL_ITER
            r = f - bag.A_sp_dict[3][0].dot(v)
            assert abs(res[0] - np.linalg.norm(r)) <= TOL_ITER * np.linalg.norm(r)
        t = orc.full_multigrid_test(orc.A_jacobi_sp_dict[3], f, True)
        for l in (1, 2):
            dev.set_rhs_true(l, bag.b_dict[l])
        dev.set_vector(3, "f", f)
        dev.fmg(2)
        got = (dev.get_vector(3, "v"), dev.get_vector(2, "f"), dev.get_vector(2, "v"), dev.get_vector(3, "err"))
        _check_fmg_tuple(got, t, f)


@pytest.mark.parametrize("dim", [2, 3])
def test_transfers_match_oracle_exactly(dim):
    bag, orc, dev = _oracle_and_device(dim, 0, 1, 6 if dim == 2 else 4, 11, 1)
    rng = np.random.default_rng(1234)
    with dev:
        xc = rng.standard_normal((bag.levels[0].n, 1))
        xf = rng.standard_normal((bag.levels[1].n, 1))
        dev.set_vector(0, "v", xc)
        dev.prolong(1, add=False)
        assert np.array_equal(dev.get_vector(1, "err"), orc.interpolate(xc, 0))
        dev.set_vector(1, "v", xf)
        dev.prolong(1, add=True)
        assert np.array_equal(dev.get_vector(1, "v"), xf + orc.interpolate(xc, 0))
        dev.set_vector(1, "r", xf)
        dev.restrict(1, "direct")
        assert np.array_equal(dev.get_vector(0, "f"), orc.restrict_direct(xf, 1))
        dev.set_vector(1, "r", xf)
        dev.restrict(1, "full_weighting")
        assert np.array_equal(dev.get_vector(0, "f"), orc.restrict_full_weighting(xf, 1))


def test_full_weighting_v_cycle_matches_oracle():
    bag, orc, dev = _oracle_and_device(2, 1, 3, 4, 5, 4)
    with dev:
        dev.set_params(4, 4, bag.omega, restriction="full_weighting")
        f = bag.b_dict[3]
        want = orc.v_cycle(orc.A_jacobi_sp_dict[3], np.zeros_like(f), f, restriction="full_weighting")
        dev.set_vector(3, "v", np.zeros_like(f))
        dev.set_vector(3, "f", f)
        dev.vcycle(3, 1)
        assert rel_l2(dev.get_vector(3, "v"), want) <= TOL_ITER


@pytest.mark.parametrize("direct", [1, 0])
def test_residual_smooth_and_coarse_solve_match_oracle(direct):
    from scipy.sparse.linalg import spsolve
    bag, orc, dev = _oracle_and_device(2, 1, 2, 8, 3, 2, coarse_direct=direct)
    rng = np.random.default_rng(7)
    with dev:
        v = rng.standard_normal((bag.levels[2].n, 1))
        f = bag.b_dict[2]
        dev.set_vector(2, "v", v)
        dev.set_vector(2, "f", f)
        dev.residual(2)
        want = f - bag.A_sp_dict[2][0].dot(v)
        assert rel_l2(dev.get_vector(2, "r"), want) <= 1e-14
        assert abs(dev.norm2(2, "r") - np.linalg.norm(want)) <= 1e-13 * np.linalg.norm(want)
        dev.set_vector(1, "f", bag.b_dict[1])
        its, rel = dev.coarse_solve()
        exact = spsolve(bag.A_sp_dict[1][0].tocsc(), bag.b_dict[1].ravel())
        assert rel <= 1e-14 and (its == 0 if direct else its > 0)      # block LU does not iterate
        assert rel_l2(dev.get_vector(1, "v"), exact) <= 1e-12
        # zero right-hand side: the solver must return zero without iterating
        dev.set_vector(1, "f", np.zeros_like(bag.b_dict[1]))
        its, rel = dev.coarse_solve()
        assert its == 0 and np.all(dev.get_vector(1, "v") == 0.0)


# ---- set-up variants must not change the arithmetic ---------------------------------------------------------------------
def _one_cycle(dev, level, f):
    dev.set_vector(level, "v", np.zeros_like(f))
    dev.set_vector(level, "f", f)
    dev.vcycle(level, 1)
    return dev.get_vector(level, "v")


@pytest.mark.parametrize("dim,lo,hi,c", [(2, 1, 3, 8), (3, 1, 3, 2)])
def test_device_generator_equals_csr_hand_off(dim, lo, hi, c):
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    bag = poisson.make_hierarchy(dim, lo, hi, c=c, mu1=3, mu2=3)
    for prune in (True, False):
        with DeviceHierarchy.from_bag(bag, dim=dim, grid_index={l: L.grid_index for l, L in bag.levels.items()},
                                      prune_zeros=prune) as a, \
                DeviceHierarchy.synthetic(dim, lo, hi, c=c, mu1=3, mu2=3, prune_zeros=prune) as b:
            for l in range(lo, hi + 1):
                ia, ib = a.level_info(l), b.level_info(l)
                assert ia == ib, (ia, ib)
                assert np.array_equal(b.get_vector(l, "f"), bag.b_dict[l])        # generated RHS is bit-identical
            va = _one_cycle(a, hi, bag.b_dict[hi])
            b.zero_vector(hi, "v")
            b.vcycle(hi, 1)
            assert np.array_equal(va, b.get_vector(hi, "v"))
            assert a.level_info(hi)["nnz_nonzero"] == int((bag.levels[hi].A.data != 0).sum())


@pytest.mark.parametrize("dim,lo,hi,c,seed", [(2, 2, 4, 8, 9), (3, 1, 3, 4, 9), (2, 2, 4, 8, None), (3, 1, 3, 4, None)])
def test_tile_shape_and_pruning_do_not_change_results(dim, lo, hi, c, seed):
    """Three matrix formats x rows per lane x block map x pruning.  The int32-column and offset-coded
    kernels add a row's entries in the order the caller stored them, the symmetric-diagonal kernel in
    ascending grid order: bit-identical for lexicographic inputs (seed None), round-off otherwise."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    bag = poisson.make_hierarchy(dim, lo, hi, c=c, mu1=4, mu2=4, seed=seed)
    gi = {l: L.grid_index for l, L in bag.levels.items()}
    f = bag.b_dict[hi]
    base = None
    for prune in (True, False):
        for R in (1, 2, 4):
            for chunk in (1, 4):
                with DeviceHierarchy.from_bag(bag, dim=dim, grid_index=gi, prune_zeros=prune, rows_per_lane=R,
                                              xcd_chunk=chunk) as dev:
                    out = _one_cycle(dev, hi, f)
                    info = dev.level_info(hi)
                    assert info["symmetric_diagonals"] in (3, 4, 8) and info["offset_codes"] == 0
                if chunk == 1:
                    others = []
                    for kw, codes in ((dict(symmetric_storage=0), (5, 7, 15)), (dict(offset_codes=0), (0,))):
                        with DeviceHierarchy.from_bag(bag, dim=dim, grid_index=gi, prune_zeros=prune, rows_per_lane=R,
                                                      xcd_chunk=chunk, **kw) as dev:
                            info = dev.level_info(hi)
                            assert info["offset_codes"] in codes and info["symmetric_diagonals"] == 0
                            others.append(_one_cycle(dev, hi, f))
                    assert np.array_equal(others[0], others[1]), (prune, R)        # same order of additions
                    if seed is None:
                        assert np.array_equal(others[0], out), (prune, R)
                    else:
                        assert rel_l2(others[0], out) <= 1e-13, (prune, R)
                # the tile shape changes which rows share a block, hence the order of the partial
                # sums in dot products; pruning removes exact zeros: equal to round-off, not bit for bit
                if base is None:
                    base = out
                else:
                    assert rel_l2(out, base) <= 1e-13, (prune, R, chunk)
                if R == 1 and chunk == 4 and prune:
                    assert np.array_equal(out, base)        # the block -> tile map alone changes nothing


# ---- size-independent properties at larger sizes ----------------------------------------------------------------------------
# 1024x1024 (5 levels); 128^3 (3 levels); BASELINE configs C2 = 2048x2048, 5 levels and C3 = 256^3, 4 levels
@pytest.mark.parametrize("dim,lo,hi", [(2, 3, 7), (3, 2, 4), (2, 4, 8), (3, 2, 5)])
def test_properties_at_scale(dim, lo, hi):
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    with DeviceHierarchy.synthetic(dim, lo, hi, c=8, mu1=2, mu2=2) as dev:
        n = dev.n_dofs(hi)
        N = dev.elements(hi)
        idx = np.arange(n)
        x = (idx % (N + 1)) / N
        y = ((idx // (N + 1)) % (N + 1)) / N
        g = 1.0 + x * x + 2.0 * y * y
        if dim == 3:
            z = (idx // (N + 1) ** 2) / N
            g = g + 3.0 * z * z
        f = dev.get_vector(hi, "f")
        # (a) the manufactured solution solves the discrete system: residual at round-off
        dev.set_vector(hi, "v", g)
        dev.residual(hi)
        assert dev.norm2(hi, "r") <= 1e-11 * np.linalg.norm(f)
        # (b) ... and is a fixed point of the V-cycle
        dev.vcycle(hi, 1)
        assert rel_l2(dev.get_vector(hi, "v"), g) <= 1e-12
        # (c) V-cycles from zero reduce the residual monotonically
        dev.zero_vector(hi, "v")
        res = dev.vcycle(hi, 3, residuals=True)
        assert res[0] < np.linalg.norm(f) and res[1] < res[0] and res[2] < res[1]
        # (d) injection of the interpolant is the identity (bit-exact)
        rng = np.random.default_rng(3)
        xc = rng.standard_normal((dev.n_dofs(hi - 1), 1))
        dev.set_vector(hi - 1, "v", xc)
        dev.prolong(hi, add=False)
        dev.copy_vector(hi, "r", "err")
        dev.restrict(hi, "direct")
        assert np.array_equal(dev.get_vector(hi - 1, "f"), xc)
        # (e) the smoother is affine: S(a) - S(b) = S0(a - b) with zero right-hand side
        a = rng.standard_normal((n, 1))
        b = rng.standard_normal((n, 1))
        outs = []
        for vec, rhs in ((a, f), (b, f), (a - b, np.zeros_like(f))):
            dev.set_vector(hi, "v", vec)
            dev.set_vector(hi, "f", rhs)
            dev.smooth(hi, 3)
            outs.append(dev.get_vector(hi, "v"))
        assert rel_l2(outs[0] - outs[1], outs[2]) <= 1e-12


# ---- edge cases and error behaviour -----------------------------------------------------------------------------------------------
def test_edge_cases_and_errors(mg):
    from multigrid_dolfinx_amd._capi import MgError
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    bag = poisson.make_hierarchy(2, 0, 1, c=4, mu1=0, mu2=0, seed=1)      # smallest grids, zero sweeps
    gi = {l: L.grid_index for l, L in bag.levels.items()}
    from oracle.mg_oracle import Oracle
    orc = Oracle(bag, gi, dim=2)
    with DeviceHierarchy.from_bag(bag, dim=2, grid_index=gi) as dev:
        f = bag.b_dict[1]
        want = orc.v_cycle(orc.A_jacobi_sp_dict[1], np.zeros_like(f), f)
        assert rel_l2(_one_cycle(dev, 1, f), want) <= TOL_ITER
        # a V-cycle entered on the coarsest level is the exact solve (multigrid.py:238-241)
        want0 = orc.v_cycle(orc.A_jacobi_sp_dict[0], np.zeros_like(bag.b_dict[0]), bag.b_dict[0])
        assert rel_l2(_one_cycle(dev, 0, bag.b_dict[0]), want0) <= 1e-12
        with pytest.raises(ValueError):
            dev.set_vector(1, "v", np.zeros(7))
        with pytest.raises(KeyError):
            dev.smooth(5, 1)
        with pytest.raises(MgError):
            dev.restrict(0, "direct")
    with pytest.raises(ValueError):
        DeviceHierarchy(2, 0, 1, c=4).set_level(1, bag.A_sp_dict[0][0])           # wrong size for the level
    bad = bag.A_sp_dict[1][0].copy()
    bad.data[bad.indices == np.repeat(np.arange(bad.shape[0]), np.diff(bad.indptr))] = 0.0
    with pytest.raises(MgError, match="diagonal"):
        DeviceHierarchy(2, 0, 1, c=4).set_level(1, bad)
    with pytest.raises(MgError, match="permutation"):
        DeviceHierarchy(2, 0, 1, c=4).set_level(1, bag.A_sp_dict[1][0], np.zeros(bag.levels[1].n, dtype=np.int64))
    with pytest.raises(RuntimeError):
        mg.A_sp_dict = None
        mg.V_cycle_scheme((None, None, 1), np.zeros((4, 1)), np.zeros((4, 1)))


def test_full_multigrid_solves_to_tolerance(mg, tmp_path, monkeypatch):
    """FullMultiGrid (multigrid.py:271-307): loop on the finest level until the residual norm is below
    the stop tolerance; the discrete solution equals the manufactured one."""
    monkeypatch.chdir(tmp_path)
    bag = poisson.make_hierarchy(2, 1, 3, c=4, mu0=2, mu1=4, mu2=4, seed=4, with_dicts=True)
    for l, a in bag.A_sp_dict.items():
        bag.A_jacobi_sp_dict[l] = mg.getJacobiMatrices(a)
    mg.configure(restriction="full_weighting", stop_tol=1e-9)
    mg.initialize_problem(bag)
    u = mg.FullMultiGrid(bag.A_jacobi_sp_dict[3], bag.b_dict[3])
    hist = bag.residual_per_V_cycle_finest
    assert len(hist) >= 1 and hist[-1] <= 1e-9 and all(h > 1e-9 for h in hist[:-1])
    assert np.abs(u - bag.levels[3].exact()).max() <= 1e-8
    rows = open(tmp_path / "iter_count_for_diff_num_elems_3_levels.csv").read().strip().split(",")
    assert rows == ["32", str(len(hist))]
    assert abs(mg.res_calculator(u, None) - np.linalg.norm(u)) <= 1e-12 * np.linalg.norm(u)


def test_traversal_order_and_streaming_loads_do_not_change_results():
    """XCD strip traversal (a different block -> slice bijection) and non-temporal loads are speed only."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    outs = []
    for strips, nt, R in ((0, 0, 2), (8, 0, 2), (8, 1, 2), (4, 1, 1), (16, 1, 4), (64, 1, 2)):
        with DeviceHierarchy.synthetic(3, 2, 4, c=8, mu1=2, mu2=2, strip_slices=strips, nontemporal=nt,
                                       rows_per_lane=R) as dev:             # 129^3 unknowns on the finest level
            dev.zero_vector(4, "v")
            res = dev.vcycle(4, 2, residuals=True)
            outs.append((dev.get_vector(4, "v"), res, R))
    for v, res, R in outs[1:]:
        if R == outs[0][2]:
            assert np.array_equal(v, outs[0][0])
        assert rel_l2(v, outs[0][0]) <= 1e-13 and np.all(np.abs(res - outs[0][1]) <= 1e-12 * outs[0][1])


def test_headline_size_checksums_agree_between_formats():
    """BASELINE config C4 at full size (1025^3 unknowns, 6 levels, one GPU).  Vectors of this size never
    cross to the host; the check is a checksum of checksums: the l2 residual norms of three V(2,2) cycles
    must fall monotonically and agree between the three matrix formats (symmetric diagonals, offset-coded
    columns with in-kernel D^-1, int32 columns with streamed D^-1), which share nothing but the arithmetic."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    norms = []
    for kw, want in ((dict(), (0, 4)), (dict(symmetric_storage=0), (7, 0)), (dict(offset_codes=0), (0, 0))):
        with DeviceHierarchy.synthetic(3, 2, 7, c=8, mu1=2, mu2=2, **kw) as dev:
            info = dev.level_info(7)
            assert info["n_global"] == 1025 ** 3 and info["ell_width"] == 7
            assert (info["offset_codes"], info["symmetric_diagonals"]) == want
            assert info["nnz_nonzero"] == 7 * 1023 ** 3 - 6 * 1023 ** 2 + (1025 ** 3 - 1023 ** 3)
            f_norm = dev.norm2(7, "f")
            dev.zero_vector(7, "v")
            res = dev.vcycle(7, 3, residuals=True)
            assert res[0] < f_norm and res[1] < res[0] and res[2] < res[1]
            norms.append(np.concatenate([[f_norm], res]))
    for other in norms[1:]:
        assert np.all(np.abs(norms[0] - other) <= 1e-12 * other)


@pytest.mark.parametrize("dim,lo,hi,c,seed", [(2, 1, 3, 8, 4), (3, 1, 3, 2, None), (3, 1, 3, 4, 6)])
def test_red_black_gauss_seidel_matches_oracle(dim, lo, hi, c, seed):
    """BASELINE config 5's smoother (no reference implementation: parity unpinned, oracle only)."""
    from multigrid_dolfinx_amd._capi import MgError
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    from oracle.mg_oracle import Oracle, rbgs_relaxation
    bag = poisson.make_hierarchy(dim, lo, hi, c=c, mu1=2, mu2=2, omega=1.0, seed=seed)
    gi = {l: L.grid_index for l, L in bag.levels.items()}
    orc = Oracle(bag, gi, dim=dim)
    rng = np.random.default_rng(5)
    f = bag.b_dict[hi]
    v0 = rng.standard_normal(f.shape)
    for R in (1, 2, 4):
        with DeviceHierarchy.from_bag(bag, dim=dim, grid_index=gi, rows_per_lane=R) as dev:
            for omega in (1.0, 1.2):
                dev.set_params(2, 2, omega, smoother="rbgs")
                dev.set_vector(hi, "v", v0)
                dev.set_vector(hi, "f", f)
                dev.smooth(hi, 3)
                want = rbgs_relaxation(bag.A_sp_dict[hi][0], v0, f, 3, omega, gi[hi] & 1)
                assert rel_l2(dev.get_vector(hi, "v"), want) <= TOL_SWEEP
            dev.set_params(2, 2, 1.0, smoother="rbgs")
            want = orc.v_cycle(orc.A_jacobi_sp_dict[hi], np.zeros_like(f), f, smoother="rbgs")
            assert rel_l2(_one_cycle(dev, hi, f), want) <= TOL_ITER
            # Gauss-Seidel smooths better than the reference's damped Jacobi at equal sweep counts
            res_gs = dev.vcycle(hi, 1, residuals=True)[0]
            dev.set_params(2, 2, 2.0 / 3.0, smoother="jacobi")
            dev.zero_vector(hi, "v")
            res_j = dev.vcycle(hi, 2, residuals=True)[1]
            assert res_gs < res_j
    # as-delivered matrices keep the same-colour (zero) couplings: the colouring is refused
    with DeviceHierarchy.from_bag(bag, dim=dim, grid_index=gi, prune_zeros=False) as dev:
        dev.set_params(1, 1, 1.0, smoother="rbgs")
        with pytest.raises(MgError, match="two-colouring"):
            dev.smooth(hi, 1)


@pytest.mark.parametrize("dim,N,seed", [(2, 16, None), (2, 32, 1), (2, 128, 2), (3, 8, 3), (3, 32, None)])
def test_direct_coarsest_solve_is_exact(dim, N, seed):
    """The block-tridiagonal LU that stands in for spsolve (multigrid.py:239-241): one block (N=16),
    several blocks of several planes (2-D) and one or two planes per block (3-D, 33^3 = BASELINE's coarsest grid);
    agrees with SuperLU and with the PCG fallback."""
    from scipy.sparse.linalg import spsolve
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    lvl = poisson.make_level(N, dim, seed=seed)
    rng = np.random.default_rng(8)
    rhs = rng.standard_normal((lvl.n, 1))
    exact = spsolve(lvl.A.tocsc(), rhs.ravel())
    sols = []
    # (direct: the default blocks of at least 2048 rows -- two planes of the 33^3 grid, sixteen lines of the 129^2 one --, one
    #  plane / line per block, round 2's blocks of 512 rows; 0: the PCG fallback)
    for direct, block_rows in ((1, None), (1, 1), (1, 512), (0, None)):
        kw = {} if block_rows is None else dict(direct_block_rows=block_rows)
        with DeviceHierarchy(dim, 0, 0, c=N, coarse_direct=direct, **kw) as dev:
            dev.set_level(0, lvl.A, lvl.grid_index)
            dev.set_params(1, 1, 2 / 3, coarse_rtol=1e-15)
            dev.set_vector(0, "f", rhs)
            its, _ = dev.coarse_solve()
            assert (its == 0) == bool(direct)
            sols.append(dev.get_vector(0, "v"))
            dev.set_vector(0, "f", lvl.b)                      # second solve re-uses the factorisation
            dev.coarse_solve()
            assert np.abs(dev.get_vector(0, "v") - lvl.exact()).max() <= 1e-11
    assert all(rel_l2(sol, exact) <= 1e-12 for sol in sols[:3]) and rel_l2(sols[3], exact) <= 1e-11


@pytest.mark.parametrize("dim,lo,hi,c", [(2, 1, 3, 8), (3, 1, 3, 4)])
def test_fused_residual_injection_is_bit_identical(dim, lo, hi, c):
    """vcycle() evaluates the residual at the coarse nodes only when restricting by injection; the coarse
    right-hand side and the iterate must equal the two-kernel path bit for bit (int32 and coded columns)."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    bag = poisson.make_hierarchy(dim, lo, hi, c=c, mu1=3, mu2=3, seed=12)
    gi = {l: L.grid_index for l, L in bag.levels.items()}
    f = bag.b_dict[hi]
    for kw in (dict(), dict(class_sweeps=0), dict(symmetric_storage=0), dict(offset_codes=0)):
        outs = []
        for fused in (1, 0):
            with DeviceHierarchy.from_bag(bag, dim=dim, grid_index=gi, **kw) as dev:
                dev.set_tuning("fuse_restrict", fused)
                v = _one_cycle(dev, hi, f)
                outs.append((v, dev.get_vector(hi - 1, "f"), dev.get_vector(hi - 1, "v")))
        for a, b in zip(*outs):
            assert np.array_equal(a, b)


def test_asymmetric_matrix_keeps_full_storage():
    """Symmetric diagonal storage stores one half of every pair, so it is only chosen for matrices whose halves agree: bit
    for bit with `storage_auto` 0 (one entry off by its last bits suffices to keep both halves), within 4 units in the last
    place by default (the automatic second try: reported by `mg_level_storage`), and never for a real asymmetry."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    bag = poisson.make_hierarchy(2, 1, 2, c=8, mu1=2, mu2=2)
    rng = np.random.default_rng(2)
    for factor, kw, want_sym in ((1.0 + 2.0 ** -50, dict(storage_auto=0), 0), (1.0 + 2.0 ** -50, dict(), 2), (1.0 + 2.0 ** -40, dict(), 0)):
        A = bag.A_sp_dict[2][0].copy()
        rows = np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))
        upper = (A.indices > rows) & (A.data != 0)
        k = np.flatnonzero(upper)[7]
        A.data[k] *= factor                                         # one entry, last bits only
        v = rng.standard_normal((A.shape[0], 1))
        with DeviceHierarchy(2, 1, 2, c=8, **kw) as dev:
            dev.set_level(1, bag.A_sp_dict[1][0], bag.levels[1].grid_index)
            dev.set_level(2, A, bag.levels[2].grid_index)
            info, st = dev.level_info(2), dev.level_storage(2)
            assert st["symmetric"] == want_sym, (factor, kw, st)
            if want_sym == 0:
                assert info["symmetric_diagonals"] == 0 and info["offset_codes"] == 5
                # the report names the pair: the lower of its two rows in lexicographic numbering (the generated level is
                # numbered that way) and its distance in units in the last place
                assert st["first_asymmetric_row"] == max(rows[k], A.indices[k]) and st["ulps_used"] == 0, st
                assert st["max_pair_ulps"] == (4 if factor < 1.0 + 2.0 ** -45 else 4096), st
            else:
                assert info["symmetric_diagonals"] == 3 and st["ulps_used"] == 4 and st["max_pair_ulps"] == 4, (info, st)
            dev.set_params(2, 2, 2 / 3)
            dev.set_vector(2, "v", v)
            dev.set_vector(2, "f", bag.b_dict[2])
            dev.residual(2)
            assert rel_l2(dev.get_vector(2, "r"), bag.b_dict[2] - A.dot(v)) <= 1e-14


@pytest.mark.parametrize("mu", [(2, 2), (3, 2), (1, 0)])
def test_graph_replay_equals_eager_launches(mu):
    """V-cycles are captured into a hipGraph and replayed; the Jacobi ping-pong state (odd sweep totals swap
    the buffers) is part of the cache key.  Five cycles must equal five eagerly launched ones bit for bit."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    bag = poisson.make_hierarchy(2, 1, 3, c=8, mu1=mu[0], mu2=mu[1], seed=7)
    gi = {l: L.grid_index for l, L in bag.levels.items()}
    f = bag.b_dict[3]
    outs = []
    for graph in (1, 0):
        with DeviceHierarchy.from_bag(bag, dim=2, grid_index=gi, graph=graph) as dev:
            dev.set_vector(3, "v", np.zeros_like(f))
            dev.set_vector(3, "f", f)
            res = []
            for _ in range(5):
                res.append(dev.vcycle(3, 1, residuals=True)[0])
            dev.set_params(mu[0] + 1, mu[1], bag.omega, keep_err=True)      # new parameters: a new capture
            dev.vcycle(3, 2)
            outs.append((dev.get_vector(3, "v"), np.array(res), dev.get_vector(3, "err"), dev.get_vector(2, "f")))
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_direct_solver_rejects_matrices_it_cannot_factor():
    """No pivoting: a coarsest matrix with a vanishing leading pivot block must fall back to PCG (or fail
    loudly), never return garbage."""
    import scipy.sparse as sps
    from multigrid_dolfinx_amd._capi import MgError
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    lvl = poisson.make_level(16, 2)
    A = lvl.A.tolil()
    interior = int(np.flatnonzero(np.abs(lvl.A.diagonal() - 4.0) < 1e-12)[0])
    A[interior, interior] = 1e-300                                  # breaks the pivot-free elimination
    A = sps.csr_matrix(A)
    with DeviceHierarchy(2, 0, 0, c=16) as dev:
        dev.set_level(0, A, lvl.grid_index)
        dev.set_params(1, 1, 2 / 3, coarse_maxit=50)
        dev.set_vector(0, "f", lvl.b)
        try:
            its, rel = dev.coarse_solve()
            assert its > 0                                          # PCG took over
            r = lvl.b - A.dot(dev.get_vector(0, "v"))
            assert np.linalg.norm(r) <= 1e-10 * np.linalg.norm(lvl.b)
        except MgError as exc:
            assert "did not converge" in str(exc)


_N128 = {}


def _n128_oracle():
    """V(50,50) on the 129^3 hierarchy by the oracle (once per session: ~half a minute of SciPy)."""
    if not _N128:
        from oracle.mg_oracle import Oracle
        bag = poisson.make_hierarchy(3, 2, 4, c=8, mu1=50, mu2=50)
        orc = Oracle(bag, {l: L.grid_index for l, L in bag.levels.items()}, dim=3)
        f = bag.b_dict[4]
        want = orc.v_cycle(orc.A_jacobi_sp_dict[4], np.zeros_like(f), f)
        r = f - bag.A_sp_dict[4][0].dot(want)
        _N128.update(want=want, res=float(np.linalg.norm(r)))
    return _N128["want"], _N128["res"]


_SMALL = dict(fuse_min_rows=0, march_min_rows=0, fuse_k_min_rows=0, fuse_k4_min_rows=0, fuse_k5_min_rows=0)


@pytest.mark.parametrize("tuning", [dict(), dict(_SMALL), dict(_SMALL, fuse_k=0), dict(_SMALL, fuse_k=3, fuse_k_shape=0),
                                    dict(_SMALL, fuse_k=5, fuse_k_shape=4), dict(_SMALL, fuse_classes=0, class_sweeps=0)])
def test_3d_n128_reference_parameters_match_oracle(tuning):
    """The largest 3-D oracle comparison that fits a test budget: 129^3 unknowns, 3 levels (coarsest 33^3 =
    BASELINE's coarsest grid), the reference's V(50,50), omega = 2/3.  3-D is parity-unpinned (no reference);
    this pins the HIP path to the CPU restatement at the north-star tolerance.  With the size thresholds at zero the
    kernels that the headline sizes run -- the K-sweep march `sdia_jacobikc`, the two-sweep passes `sdia_jacobi2c` /
    `sdia_jacobi2p`, the one-sweep march `sdia_sweep1c` -- face the oracle themselves, not only the slice kernels they
    are bit-identical to (default thresholds: 129^3 rows stay below them)."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    want, res_want = _n128_oracle()
    with DeviceHierarchy.synthetic(3, 2, 4, c=8, mu1=50, mu2=50, **tuning) as dev:
        if tuning:
            assert dev.time_kernel("jacobi2", 4, 1) > 0             # the smoother does run the march kernels on this level
        dev.zero_vector(4, "v")
        res = dev.vcycle(4, 1, residuals=True)
        got = dev.get_vector(4, "v")
    assert rel_l2(got, want) <= TOL_ITER
    assert abs(res[0] - res_want) <= TOL_ITER * res_want


def test_p2_plane_march_faces_the_oracle_directly():
    """BASELINE config 5's kernels at a size the oracle still does in seconds: the 65^3-point P2 lattice (3 levels, nine-colour
    Gauss-Seidel V(2,2), P2 transfer pair) with `lattice_march_min_rows` 0, so that `lat_march` -- not only the gathering
    kernel it is bit-identical to -- is compared with oracle/mg_oracle.py at the north-star tolerance.  No reference
    implementation exists for P2 / Gauss-Seidel: parity unpinned, the oracle is the only independent check."""
    import types
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    from oracle.mg_oracle import Oracle
    dim, lo, hi, c = 3, 1, 3, 8                        # lattices of 16, 32, 64 steps per dimension
    levels = {l: poisson.p2_level(c * 2 ** l // 2, dim) for l in range(lo, hi + 1)}
    bag = types.SimpleNamespace(
        mesh_dof_list_dict={}, element_size={l: 1.0 / L.N for l, L in levels.items()}, coarsest_level_elements_per_dim=c,
        coarsest_level=lo, finest_level=hi, A_sp_dict={l: (L.A, l) for l, L in levels.items()}, A_jacobi_sp_dict={},
        b_dict={l: L.b for l, L in levels.items()}, mu0=1, mu1=2, mu2=2, omega=1.0,
        residual_per_V_cycle_finest=[], error_per_V_cycle_finest=[], u_exact_fine=None, V_fine_dolfx=None, levels=levels)
    orc = Oracle(bag, {l: L.grid_index for l, L in levels.items()}, dim=dim)
    orc.prolongation_table = poisson.p2_prolongation_table(dim)
    orc.restriction_table = poisson.p2_restriction_table(dim)
    f = bag.b_dict[hi]
    want = orc.v_cycle(orc.A_jacobi_sp_dict[hi], np.zeros_like(f), f, smoother="mcgs", restriction="table")
    r_want = float(np.linalg.norm(f - bag.A_sp_dict[hi][0].dot(want)))
    for march in (1, 0):
        with DeviceHierarchy.synthetic_p2(dim, lo, hi, c=c, mu1=2, mu2=2, omega=1.0, transfers="p2", restriction="table",
                                          lattice_march=march, lattice_march_min_rows=0) as dev:
            dev.zero_vector(hi, "v")
            res = dev.vcycle(hi, 1, residuals=True)
            got = dev.get_vector(hi, "v")
        assert rel_l2(got, want) <= TOL_ITER, march
        assert abs(res[0] - r_want) <= TOL_ITER * r_want, march


@pytest.mark.parametrize("dim,c,seed", [(2, 5, None), (2, 7, 3), (3, 3, None), (3, 5, 8), (2, 1, None), (3, 1, 2)])
def test_unusual_grid_sizes_match_oracle(dim, c, seed):
    """Coarsest grids that are not powers of two (odd node counts per axis, tiny grids with c = 1): the
    slice / strip / lead-region arithmetic must not depend on friendly sizes."""
    bag, orc, dev = _oracle_and_device(dim, 1, 3, c, seed, 3)
    with dev:
        f = bag.b_dict[3]
        want = orc.v_cycle(orc.A_jacobi_sp_dict[3], np.zeros_like(f), f)
        got = _one_cycle(dev, 3, f)
        assert rel_l2(got, want) <= TOL_ITER
        t = orc.full_multigrid_test(orc.A_jacobi_sp_dict[3], f, True)
        for l in (1, 2):
            dev.set_rhs_true(l, bag.b_dict[l])
        dev.set_vector(3, "f", f)
        dev.fmg(2)
        assert rel_l2(dev.get_vector(3, "v"), t[0]) <= TOL_ITER


def test_mass_matrix_norm_and_error_calculators(mg):
    """`res_calculator` / `err_calculator` (multigrid.py:203-218) assemble sqrt(int r^2) with dolfinx; here the
    caller passes the P1 mass matrix M and the device evaluates sqrt(r^T M r) (SpMV fused with the dot)."""
    import scipy.sparse as sps
    N = 24
    n1 = N + 1
    h = 1.0 / N
    # P1 mass matrix of the right-diagonal unit-square mesh: h^2/12 * (6 on the diagonal, 1 per edge), halved /
    # quartered weights on the boundary follow from summing element contributions
    idx = np.arange(n1 * n1).reshape(n1, n1)
    rows, cols, vals = [], [], []
    loc = (h * h / 24.0) * (np.ones((3, 3)) + np.eye(3))
    for j in range(N):
        for i in range(N):
            for tri in ((idx[j, i], idx[j, i + 1], idx[j + 1, i + 1]), (idx[j, i], idx[j + 1, i + 1], idx[j + 1, i])):
                for a in range(3):
                    for b in range(3):
                        rows.append(tri[a]); cols.append(tri[b]); vals.append(loc[a, b])
    M = sps.csr_matrix((vals, (rows, cols)), shape=(n1 * n1, n1 * n1))
    M.sum_duplicates()
    rng = np.random.default_rng(4)
    r = rng.standard_normal((n1 * n1, 1))
    u = rng.standard_normal((n1 * n1, 1))
    want = float(np.sqrt((r.T @ (M @ r)).item()))
    assert abs(mg.res_calculator(r, M) - want) <= 1e-13 * want
    assert abs(mg.res_calculator(r, None) - np.linalg.norm(r)) <= 1e-13 * np.linalg.norm(r)
    d = u - r
    want = float(np.sqrt((d.T @ (M @ d)).item()))
    assert abs(mg.err_calculator(u, r, M) - want) <= 1e-13 * want
    ones = np.ones((n1 * n1, 1))
    assert abs(mg.res_calculator(ones, M) - 1.0) <= 1e-13          # sqrt(area of the unit square)


@pytest.mark.parametrize("dim,cells,seed", [(2, (4, 8, 16), None), (2, (4, 8, 16), 6), (3, (2, 4, 8), None), (3, (2, 4, 8), 1)])
def test_p2_matrices_wide_stencils_match_oracle(dim, cells, seed):
    """BASELINE config 5's element type (no reference implementation: parity unpinned).  P2 stiffness matrices
    live on the (2N+1)^dim lattice, reach two lattice planes and have up to 51 entries per row, so they take
    the run-time-width offset-coded kernels (not bit-for-bit symmetric after assembly) and -- where
    one plane per block would not be block tridiagonal -- the PCG coarsest solve.  Transfers are the lattice
    injection / Q1 interpolation."""
    import types
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    from oracle.mg_oracle import Oracle
    levels = {l: poisson.p2_level(N, dim, seed=seed) for l, N in enumerate(cells)}
    c = 2 * cells[0]
    bag = types.SimpleNamespace(
        mesh_dof_list_dict={}, element_size={l: 1.0 / L.N for l, L in levels.items()}, coarsest_level_elements_per_dim=c,
        coarsest_level=0, finest_level=2, A_sp_dict={l: (L.A, l) for l, L in levels.items()}, A_jacobi_sp_dict={},
        b_dict={l: L.b for l, L in levels.items()}, mu0=2, mu1=3, mu2=3, omega=0.5,
        residual_per_V_cycle_finest=[], error_per_V_cycle_finest=[], u_exact_fine=None, V_fine_dolfx=None)
    gi = {l: L.grid_index for l, L in levels.items()}
    orc = Oracle(bag, gi, dim=dim)
    f = bag.b_dict[2]
    want = orc.v_cycle(orc.A_jacobi_sp_dict[2], np.zeros_like(f), f)
    with DeviceHierarchy.from_bag(bag, dim=dim, grid_index=gi) as dev:
        info = dev.level_info(2)
        # 3-D: up to 51 entries per row, ~100 distinct offsets -> run-time-width offset codes; the 2-D matrix
        # (9 per row, 13 offsets, bit-for-bit symmetric when lexicographic) may take the symmetric diagonals
        assert info["ell_width"] > 7 and (info["offset_codes"] > 7 or info["symmetric_diagonals"] == 8)
        if dim == 3:
            assert info["offset_codes"] > 15 and info["symmetric_diagonals"] == 0
        got = _one_cycle(dev, 2, f)
        assert rel_l2(got, want) <= TOL_ITER
        # P2 reproduces the quadratic manufactured solution: it is a fixed point of the cycle
        exact = levels[2].exact()
        dev.set_vector(2, "v", exact)
        dev.residual(2)
        assert dev.norm2(2, "r") <= 1e-11 * np.linalg.norm(f)
        dev.vcycle(2, 1)
        assert rel_l2(dev.get_vector(2, "v"), exact) <= 1e-11


@pytest.mark.parametrize("dim,cells,seed", [(2, (4, 8, 16), 6), (3, (2, 4, 8), None), (3, (4, 8, 16), 1)])
def test_nine_colour_gauss_seidel_on_p2_matches_oracle(dim, cells, seed):
    """BASELINE config 5: P2 rows + Gauss-Seidel (no reference implementation: parity unpinned, oracle only).  P2
    rows couple unknowns of equal index parity, so the smoother uses nine lattice colours (`smoother="mcgs"`); red-black
    is refused.  Sweeps and whole cycles against the oracle (33^3 lattice in the last case), the manufactured solution as
    a fixed point, residuals falling monotonically."""
    import types
    from multigrid_dolfinx_amd._capi import MgError
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    from oracle.mg_oracle import Oracle, lattice9_colors, rbgs_relaxation
    levels = {l: poisson.p2_level(N, dim, seed=seed) for l, N in enumerate(cells)}
    c = 2 * cells[0]
    bag = types.SimpleNamespace(
        mesh_dof_list_dict={}, element_size={l: 1.0 / L.N for l, L in levels.items()}, coarsest_level_elements_per_dim=c,
        coarsest_level=0, finest_level=2, A_sp_dict={l: (L.A, l) for l, L in levels.items()}, A_jacobi_sp_dict={},
        b_dict={l: L.b for l, L in levels.items()}, mu0=2, mu1=2, mu2=2, omega=1.0,
        residual_per_V_cycle_finest=[], error_per_V_cycle_finest=[], u_exact_fine=None, V_fine_dolfx=None)
    gi = {l: L.grid_index for l, L in levels.items()}
    orc = Oracle(bag, gi, dim=dim)
    f = bag.b_dict[2]
    rng = np.random.default_rng(9)
    v0 = rng.standard_normal(f.shape)
    A = bag.A_sp_dict[2][0].copy()
    A.eliminate_zeros()
    color = lattice9_colors(gi[2], levels[2].N, dim)
    for R in (1, 2):
        with DeviceHierarchy.from_bag(bag, dim=dim, grid_index=gi, rows_per_lane=R) as dev:
            for omega in (1.0, 1.15):
                dev.set_params(2, 2, omega, smoother="mcgs")
                dev.set_vector(2, "v", v0)
                dev.set_vector(2, "f", f)
                dev.smooth(2, 2)
                want = rbgs_relaxation(A, v0, f, 2, omega, color)
                assert rel_l2(dev.get_vector(2, "v"), want) <= TOL_SWEEP
            dev.set_params(2, 2, 1.0, smoother="mcgs")
            want = orc.v_cycle(orc.A_jacobi_sp_dict[2], np.zeros_like(f), f, smoother="mcgs")
            assert rel_l2(_one_cycle(dev, 2, f), want) <= TOL_ITER
            res = dev.vcycle(2, 6, residuals=True)
            assert np.all(res[1:] < res[:-1])
            exact = levels[2].exact()
            dev.set_vector(2, "v", exact)
            dev.vcycle(2, 1)
            assert rel_l2(dev.get_vector(2, "v"), exact) <= 1e-11
            dev.set_params(1, 1, 1.0, smoother="rbgs")
            with pytest.raises(MgError, match="two-colouring"):
                dev.smooth(2, 1)


@pytest.mark.parametrize("dim,c,lo,hi", [(2, 4, 0, 3), (3, 4, 0, 2)])
def test_device_p2_generator_equals_the_assembled_levels(dim, c, lo, hi):
    """`mg_gen_lattice_level` writes P2 levels from the eight per-class interior stencils (no host matrix: the only way
    to set up the 513^3-point lattice of BASELINE config 5).  It must be the matrix `poisson.p2_level` assembles: the
    same residuals and sweeps on random vectors (to round-off: the assembly sums element contributions in its own
    order), the same right-hand side, the same cycles."""
    import types
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    levels = {l: poisson.p2_level(c * 2 ** l // 2, dim) for l in range(lo, hi + 1)}
    bag = types.SimpleNamespace(
        mesh_dof_list_dict={}, element_size={l: 1.0 / L.N for l, L in levels.items()}, coarsest_level_elements_per_dim=c,
        coarsest_level=lo, finest_level=hi, A_sp_dict={l: (L.A, l) for l, L in levels.items()}, A_jacobi_sp_dict={},
        b_dict={l: L.b for l, L in levels.items()}, mu0=2, mu1=2, mu2=2, omega=1.0,
        residual_per_V_cycle_finest=[], error_per_V_cycle_finest=[], u_exact_fine=None, V_fine_dolfx=None)
    gi = {l: L.grid_index for l, L in levels.items()}
    rng = np.random.default_rng(12)
    with DeviceHierarchy.from_bag(bag, dim=dim, grid_index=gi) as a, \
            DeviceHierarchy.synthetic_p2(dim, lo, hi, c=c, mu1=2, mu2=2, omega=1.0) as b:
        for l in range(lo, hi + 1):
            ia, ib = a.level_info(l), b.level_info(l)
            assert ia["n_global"] == ib["n_global"] and ia["nnz_nonzero"] == ib["nnz_nonzero"]
            fb = b.get_vector(l, "f")
            assert np.abs(fb - bag.b_dict[l]).max() <= 1e-13 * np.abs(bag.b_dict[l]).max()
            if l == lo:
                continue
            v = rng.standard_normal(fb.shape)
            for dev in (a, b):
                dev.set_params(2, 2, 2.0 / 3.0, smoother="jacobi")
                dev.set_vector(l, "v", v)
                dev.set_vector(l, "f", bag.b_dict[l])
            a.residual(l); b.residual(l)
            assert rel_l2(b.get_vector(l, "r"), a.get_vector(l, "r")) <= 1e-14, l
            a.smooth(l, 2); b.smooth(l, 2)
            assert rel_l2(b.get_vector(l, "v"), a.get_vector(l, "v")) <= 1e-14, l
        for dev in (a, b):
            dev.set_params(2, 2, 1.0, smoother="mcgs")
            dev.set_vector(hi, "f", bag.b_dict[hi])
            dev.zero_vector(hi, "v")
        ra, rb = a.vcycle(hi, 3, residuals=True), b.vcycle(hi, 3, residuals=True)
        assert np.all(np.abs(ra - rb) <= 1e-12 * ra)
        assert rb[-1] < rb[0]


@pytest.mark.parametrize("dim,cells,seed", [(2, (4, 8, 16), 6), (3, (2, 4, 8), None), (3, (4, 8, 16), 1)])
def test_p2_table_prolongation_matches_oracle(dim, cells, seed):
    """`mg_set_prolongation_table` with the natural P2 embedding (BASELINE config 5's transfer operator; no reference:
    parity unpinned, the oracle's table prolongation is pinned to quadratic reproduction): bit-exact against the oracle,
    whole cycles to the north-star tolerance, residuals fall monotonically, and switching back restores the bilinear
    table's cycles.  (With the reference's injection of the finite-element residual -- SURVEY.md App. A Q1 -- the coarse
    correction is under-scaled either way, so the prolongation alone changes the rate by a per cent or so.)"""
    import types
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    from oracle.mg_oracle import Oracle
    levels = {l: poisson.p2_level(N, dim, seed=seed) for l, N in enumerate(cells)}
    c = 2 * cells[0]
    bag = types.SimpleNamespace(
        mesh_dof_list_dict={}, element_size={l: 1.0 / L.N for l, L in levels.items()}, coarsest_level_elements_per_dim=c,
        coarsest_level=0, finest_level=2, A_sp_dict={l: (L.A, l) for l, L in levels.items()}, A_jacobi_sp_dict={},
        b_dict={l: L.b for l, L in levels.items()}, mu0=2, mu1=2, mu2=2, omega=1.0,
        residual_per_V_cycle_finest=[], error_per_V_cycle_finest=[], u_exact_fine=None, V_fine_dolfx=None)
    gi = {l: L.grid_index for l, L in levels.items()}
    orc = Oracle(bag, gi, dim=dim)
    table = poisson.p2_prolongation_table(dim)
    rng = np.random.default_rng(2)
    f = bag.b_dict[2]
    with DeviceHierarchy.from_bag(bag, dim=dim, grid_index=gi) as dev:
        dev.set_params(2, 2, 1.0, smoother="mcgs")
        dev.set_vector(2, "f", f)
        dev.zero_vector(2, "v")
        res_q1 = dev.vcycle(2, 4, residuals=True)
        dev.set_prolongation("p2")
        for l in (0, 1):
            vc = rng.standard_normal((levels[l].n, 1))
            dev.set_vector(l, "v", vc)
            dev.prolong(l + 1, add=False)
            assert np.array_equal(dev.get_vector(l + 1, "err"), orc.interpolate_table(vc, l, table)), l
        orc.prolongation_table = table
        want = orc.v_cycle(orc.A_jacobi_sp_dict[2], np.zeros_like(f), f, smoother="mcgs")
        assert rel_l2(_one_cycle(dev, 2, f), want) <= TOL_ITER
        dev.zero_vector(2, "v")
        res_p2 = dev.vcycle(2, 4, residuals=True)
        assert np.all(res_p2[1:] < res_p2[:-1]) and res_p2[-1] < 1.05 * res_q1[-1], (res_q1, res_p2)
        exact = levels[2].exact()
        dev.set_vector(2, "v", exact)
        dev.vcycle(2, 1)
        assert rel_l2(dev.get_vector(2, "v"), exact) <= 1e-11
        # ... and with the transpose of that prolongation as restriction (the canonical finite-element pair) the cycle
        # becomes a textbook multigrid: bit-exact transfers, cycles to tolerance, a much better rate than injection
        rtab = poisson.p2_restriction_table(dim)
        for l in (1, 2):
            r = rng.standard_normal((levels[l].n, 1))
            dev.set_vector(l, "r", r)
            dev.restrict(l, "table")
            assert np.array_equal(dev.get_vector(l - 1, "f"), orc.restrict_table(r, l, rtab)), l
        orc.restriction_table = rtab
        dev.set_params(2, 2, 1.0, smoother="mcgs", restriction="table")
        want = orc.v_cycle(orc.A_jacobi_sp_dict[2], np.zeros_like(f), f, restriction="table", smoother="mcgs")
        assert rel_l2(_one_cycle(dev, 2, f), want) <= TOL_ITER
        dev.zero_vector(2, "v")
        res_fe = dev.vcycle(2, 4, residuals=True)
        assert np.all(res_fe[1:] < 0.35 * res_fe[:-1]) and res_fe[-1] < 1e-2 * res_p2[-1], (res_p2, res_fe)
        dev.set_vector(2, "v", exact)
        dev.vcycle(2, 1)
        assert rel_l2(dev.get_vector(2, "v"), exact) <= 1e-11
        dev.set_prolongation("q1")
        dev.set_params(2, 2, 1.0, smoother="mcgs")
        dev.zero_vector(2, "v")
        assert np.all(np.abs(dev.vcycle(2, 4, residuals=True) - res_q1) <= 1e-12 * res_q1)


@pytest.mark.parametrize("c,lo,hi,tile,gs2", [(4, 1, 3, 1, 0), (8, 1, 3, 1, 0), (6, 1, 3, 1, 0), (4, 1, 3, 2, 0), (8, 1, 3, 2, 0),
                                              (6, 1, 3, 2, 0), (8, 2, 4, 0, 0), (4, 1, 3, 0, 1), (8, 1, 3, 0, 1), (6, 1, 3, 0, 1),
                                              (8, 2, 4, 0, 1)])
def test_lattice_plane_march_equals_the_gathering_kernel(c, lo, hi, tile, gs2):
    """P2 levels as a plane march with x in LDS (`lat_march`, mg_lattice.hip.h) against the kernel it replaces
    (`ell_cls_apply`, every neighbour gathered from global memory): residual, weighted-Jacobi sweeps, every colour of the
    nine-colour Gauss-Seidel sweep and whole cycles, bit for bit -- both apply a row's entries in stored order.  Lattices
    of 17^3 .. 129^3 points, both tile shapes (64 x 16 and 128 x 16 cells): tiles that stick out of the grid on every side,
    boundary classes in every wave, a level size (6 * 8 + 1) that is no multiple of the tile.  `gs2`: the Gauss-Seidel sweep
    with two colours per launch, out of place (`lat_gs2`) -- five launches instead of nine, the same bits."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    rng = np.random.default_rng(31)
    with DeviceHierarchy.synthetic_p2(3, lo, hi, c=c, mu1=2, mu2=2, omega=1.0, lattice_march=0) as ref, \
            DeviceHierarchy.synthetic_p2(3, lo, hi, c=c, mu1=2, mu2=2, omega=1.0, lattice_march=1,
                                         lattice_march_min_rows=0, lattice_segments=2, lattice_tile=tile, lattice_gs2=gs2) as new:
        n = ref.level_info(hi)["n_global"]
        v = rng.standard_normal((n, 1))
        f = rng.standard_normal((n, 1))
        out = {}
        for name, dev in (("ref", ref), ("new", new)):
            got = []
            dev.set_vector(hi, "f", f)
            dev.set_vector(hi, "v", v)
            dev.residual(hi)
            got.append(dev.get_vector(hi, "r"))
            dev.set_params(3, 3, 0.6, smoother="jacobi")
            dev.smooth(hi, 3)
            got.append(dev.get_vector(hi, "v"))
            dev.set_params(2, 2, 1.0, smoother="mcgs")
            dev.set_vector(hi, "v", v)
            dev.smooth(hi, 2)
            got.append(dev.get_vector(hi, "v"))
            dev.zero_vector(hi, "v")
            got.append(np.asarray(dev.vcycle(hi, 2, residuals=True)))
            got.append(dev.get_vector(hi, "v"))
            out[name] = got
        for k, (a, b) in enumerate(zip(out["ref"], out["new"])):
            assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))
        assert new.time_kernel("gs", hi, 1) > 0.0


def test_config5_full_size_properties():
    """BASELINE config 5 at its full size on one GPU (P2 on the 513^3-point lattice, 135 M unknowns, nine-colour
    Gauss-Seidel; no reference and no oracle at this size): properties that do not depend on the size -- the quadratic
    manufactured solution is reproduced by P2 (zero residual, fixed point of a cycle), residual norms fall monotonically
    from a zero guess, a Gauss-Seidel sweep is deterministic."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    with DeviceHierarchy.synthetic_p2(3, 2, 6, c=8, mu1=2, mu2=2, omega=1.0) as dev:
        info = dev.level_info(6)
        n1 = 513
        assert info["n_global"] == n1 ** 3 and info["ell_width"] == 51
        t = np.arange(n1, dtype=np.float64) / (n1 - 1)
        exact = (1.0 + t[None, None, :] ** 2 + 2.0 * t[None, :, None] ** 2 + 3.0 * t[:, None, None] ** 2).reshape(-1, 1)
        f_norm = dev.norm2(6, "f")
        dev.set_vector(6, "v", exact)
        dev.residual(6)
        assert dev.norm2(6, "r") <= 1e-10 * f_norm
        dev.vcycle(6, 1)
        got = dev.get_vector(6, "v")
        assert rel_l2(got, exact) <= 1e-11
        del got, exact
        dev.zero_vector(6, "v")
        res = dev.vcycle(6, 4, residuals=True)
        assert np.all(res[1:] < res[:-1]) and res[-1] < 0.5 * res[0], res
        a = dev.get_vector(6, "v")
        dev.zero_vector(6, "v")
        dev.vcycle(6, 4)
        assert np.array_equal(a, dev.get_vector(6, "v"))


def test_full_multigrid_with_the_reference_norms(mg, tmp_path, monkeypatch):
    """FullMultiGrid's stop test in the reference's own terms (multigrid.py:288-302): residual and error in the
    L2(Omega) norm, here through the P1 mass matrix handed over in the `V_fine_dolfx` slot."""
    import scipy.sparse as sps
    monkeypatch.chdir(tmp_path)
    bag = poisson.make_hierarchy(2, 1, 3, c=4, mu0=2, mu1=4, mu2=4, seed=2, with_dicts=True)
    for l, a in bag.A_sp_dict.items():
        bag.A_jacobi_sp_dict[l] = mg.getJacobiMatrices(a)
    N = 32
    h = 1.0 / N
    lvl = bag.levels[3]
    ij = np.rint(lvl.coords[:, :2] * N).astype(int)
    node = {(i, j): d for d, (i, j) in enumerate(map(tuple, ij))}
    rows, cols, vals = [], [], []
    loc = (h * h / 24.0) * (np.ones((3, 3)) + np.eye(3))
    for j in range(N):
        for i in range(N):
            for tri in (((i, j), (i + 1, j), (i + 1, j + 1)), ((i, j), (i + 1, j + 1), (i, j + 1))):
                t = [node[p] for p in tri]
                for a in range(3):
                    for b in range(3):
                        rows.append(t[a]); cols.append(t[b]); vals.append(loc[a, b])
    M = sps.csr_matrix((vals, (rows, cols)), shape=(lvl.n, lvl.n))
    M.sum_duplicates()
    bag.V_fine_dolfx = M
    bag.u_exact_fine = lvl.exact()
    mg.configure(restriction="full_weighting", stop_tol=1e-9)
    mg.initialize_problem(bag)
    u = mg.FullMultiGrid(bag.A_jacobi_sp_dict[3], bag.b_dict[3])
    res, err = bag.residual_per_V_cycle_finest, bag.error_per_V_cycle_finest
    assert len(res) == len(err) >= 1 and res[-1] <= 1e-9 and all(r > 1e-9 for r in res[:-1])
    r = bag.b_dict[3] - bag.A_sp_dict[3][0].dot(u)
    assert abs(res[-1] - float(np.sqrt((r.T @ (M @ r)).item()))) <= 1e-6 * res[-1] + 1e-15
    e = u - lvl.exact()
    assert abs(err[-1] - float(np.sqrt((e.T @ (M @ e)).item()))) <= 1e-9
    assert err[-1] < err[0]


def _p1_mass_matrix(lvl, N):
    """P1 mass matrix of the right-diagonal unit-square mesh in the level's DoF numbering."""
    import scipy.sparse as sps
    h = 1.0 / N
    ij = np.rint(lvl.coords[:, :2] * N).astype(int)
    node = {(i, j): d for d, (i, j) in enumerate(map(tuple, ij))}
    rows, cols, vals = [], [], []
    loc = (h * h / 24.0) * (np.ones((3, 3)) + np.eye(3))
    for j in range(N):
        for i in range(N):
            for tri in (((i, j), (i + 1, j), (i + 1, j + 1)), ((i, j), (i + 1, j + 1), (i, j + 1))):
                t = [node[p] for p in tri]
                for a in range(3):
                    for b in range(3):
                        rows.append(t[a]); cols.append(t[b]); vals.append(loc[a, b])
    M = sps.csr_matrix((vals, (rows, cols)), shape=(lvl.n, lvl.n))
    M.sum_duplicates()
    return M


def test_full_multigrid_keeps_its_norms_on_the_device(mg, tmp_path, monkeypatch):
    """multigrid.py:288-302 on the device (mg_fmg_ex): however many cycles the stop test takes, the same number of
    whole vectors crosses PCIe (hand-over of b_dict / f / u_exact, the result) -- per cycle only two doubles do."""
    monkeypatch.chdir(tmp_path)
    counts = []
    for tol in (1e-4, 1e-9):
        bag = poisson.make_hierarchy(2, 1, 3, c=4, mu0=2, mu1=4, mu2=4, seed=2, with_dicts=True)
        for l, a in bag.A_sp_dict.items():
            bag.A_jacobi_sp_dict[l] = mg.getJacobiMatrices(a)
        bag.V_fine_dolfx = _p1_mass_matrix(bag.levels[3], 32)
        bag.u_exact_fine = bag.levels[3].exact()
        mg.configure(restriction="full_weighting", stop_tol=tol)
        mg.initialize_problem(bag)
        mg.FullMultiGrid(bag.A_jacobi_sp_dict[3], bag.b_dict[3])
        c = mg._hierarchy().counters()
        counts.append((len(bag.residual_per_V_cycle_finest), c["uploads"], c["downloads"]))
        assert len(bag.error_per_V_cycle_finest) == len(bag.residual_per_V_cycle_finest)
    (n1, up1, down1), (n2, up2, down2) = counts
    assert n2 > n1 + 2, counts
    assert (up1, down1) == (up2, down2), counts
    assert up1 <= 6 and down1 <= 2, counts


def test_mass_matrix_can_move_to_another_level():
    """`mg_set_mass_csr` on level 2, FMG in the mass norm, then on the (four times larger) level 3 and FMG again: the
    work vector of the mass form must follow the level (it used to keep the first level's size: an out-of-bounds
    device write through the public C ABI)."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    bag = poisson.make_hierarchy(2, 1, 3, c=4, mu0=2, mu1=4, mu2=4, seed=2)
    gi = {l: L.grid_index for l, L in bag.levels.items()}
    with DeviceHierarchy.from_bag(bag, dim=2, grid_index=gi) as dev:
        for top, N in ((2, 16), (3, 32), (2, 16)):
            lvl = bag.levels[top]
            M = _p1_mass_matrix(lvl, N)
            dev.set_mass(top, M)
            dev.set_exact(top, lvl.exact())
            for l in range(1, top):
                dev.set_rhs_true(l, bag.b_dict[l])
            dev.set_vector(top, "f", bag.b_dict[top])
            res, err = dev.fmg(2, top_level=top, norm="mass", errors=True)
            u = dev.get_vector(top, "v")
            r = bag.b_dict[top] - bag.A_sp_dict[top][0].dot(u)
            e = u - lvl.exact()
            want_r = float(np.sqrt((r.T @ (M @ r)).item()))
            want_e = float(np.sqrt((e.T @ (M @ e)).item()))
            assert abs(res[-1] - want_r) <= 1e-9 * want_r + 1e-15
            assert abs(err[-1] - want_e) <= 1e-9 * want_e + 1e-15
        # a level that is set again drops what the handle kept for its old geometry
        dev.set_level(2, bag.A_sp_dict[2][0], gi[2])
        with pytest.raises(Exception):
            dev.fmg(2, top_level=2, norm="mass")


def test_full_multigrid_never_changes_norm_silently(mg, tmp_path, monkeypatch):
    """A `V_fine_dolfx` that is neither None nor a SciPy mass matrix is a dolfinx function space in the reference
    (multigrid.py:292-296).  Without dolfinx its mass matrix cannot be assembled: that is an error, not a switch to
    the l2 norm (whose 1e-11 is a different stop test).  `configure(norm="l2")` asks for the l2 norm explicitly, and
    then the error history is still written every cycle."""
    monkeypatch.chdir(tmp_path)
    bag = poisson.make_hierarchy(2, 1, 3, c=4, mu0=2, mu1=4, mu2=4, seed=2, with_dicts=True)
    for l, a in bag.A_sp_dict.items():
        bag.A_jacobi_sp_dict[l] = mg.getJacobiMatrices(a)
    bag.V_fine_dolfx = object()
    bag.u_exact_fine = bag.levels[3].exact()
    mg.configure(restriction="full_weighting", stop_tol=1e-8)
    mg.initialize_problem(bag)
    with pytest.raises(TypeError):
        mg.FullMultiGrid(bag.A_jacobi_sp_dict[3], bag.b_dict[3])
    mg.configure(restriction="full_weighting", stop_tol=1e-8, norm="l2")
    mg.initialize_problem(bag)
    u = mg.FullMultiGrid(bag.A_jacobi_sp_dict[3], bag.b_dict[3])
    res, err = bag.residual_per_V_cycle_finest, bag.error_per_V_cycle_finest
    assert len(res) == len(err) >= 1 and res[-1] <= 1e-8
    r = bag.b_dict[3] - bag.A_sp_dict[3][0].dot(u)
    assert abs(res[-1] - np.linalg.norm(r)) <= 1e-6 * res[-1] + 1e-15
    assert abs(err[-1] - np.linalg.norm(u - bag.levels[3].exact())) <= 1e-9


def test_shim_replays_the_captured_cycle(mg):
    """`V_cycle_scheme` through the drop-in module sets the same parameters before every call: that must not
    invalidate the captured V-cycle (mg_set_params only counts real changes), so the second call replays it."""
    g, bag = _init_from_fixture(mg, "c1_lex")
    f = bag.b_dict[3]
    v = np.zeros_like(f)
    for _ in range(3):
        v = mg.V_cycle_scheme(bag.A_jacobi_sp_dict[3], v, f)
    c = mg._hierarchy().counters()
    assert c["graphs_cached"] == 1 and c["graph_replays"] == 2, c
    mg._hierarchy().set_params(50, 50, 0.5)                 # a real change: new capture
    mg._hier_params = None
    mg.V_cycle_scheme(bag.A_jacobi_sp_dict[3], v, f)
    assert mg._hierarchy().counters()["graphs_cached"] >= 1


def test_prepare_cycle_only_moves_the_lazy_set_up_forward():
    """`mg_prepare_cycle` builds what the first V-cycle would build lazily (direct coarsest solve and its validation,
    colouring checks, work vectors): calling it -- once or twice, or not at all -- leaves the cycles bit for bit alike, and
    the first cycle after it is already the captured one."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    out = []
    for prepare in (0, 1, 2):
        with DeviceHierarchy.synthetic(3, 1, 4, c=4, mu1=3, mu2=3) as h:
            for _ in range(prepare):
                h.prepare_cycle(4)
            h.zero_vector(4, "v")
            res = h.vcycle(4, 3, residuals=True)
            out.append((np.asarray(res), h.get_vector(4, "v"), h.counters()))
    for res, v, counters in out[1:]:
        assert np.array_equal(res, out[0][0]) and np.array_equal(v, out[0][1])
        assert counters["graphs_cached"] >= 1 and counters["graph_replays"] >= 1, counters


def test_adhoc_contexts_are_bounded(mg):
    """Stand-alone contexts (norms of vectors of many lengths, transfers between many mesh pairs) live in one
    bounded LRU cache; `configure` / `initialize_problem` drop them."""
    rng = np.random.default_rng(0)
    for n in range(40, 60):
        r = rng.standard_normal((n, 1))
        assert abs(mg.res_calculator(r, None) - np.linalg.norm(r)) <= 1e-13 * np.linalg.norm(r)
    assert len(mg._adhoc) <= 8
    mg.configure(dim=2)
    assert len(mg._adhoc) == 0 and len(mg._grid_cache) == 0


@pytest.mark.parametrize("c,lo,hi", [(8, 2, 4), (5, 1, 3), (7, 1, 4)])
def test_two_sweep_kernel_is_bit_identical_to_single_sweeps(c, lo, hi):
    """mg_jacobi2.hip.h: two Jacobi sweeps per pass over the matrix (tile + plane march, intermediate iterate in
    LDS / registers) must reproduce two launches of the one-sweep kernel bit for bit -- for every tile shape,
    plane segmentation, odd sweep count and grid size that is not a multiple of the tile."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    rng = np.random.default_rng(c)
    want = {}
    # (fuse_k=0: pairs of sweeps only, the two-sweep pass; fuse_k=3..5: the K-sweep march of mg_jacobik3d.hip.h in its
    #  three tile shapes, with and without the DPP neighbour exchange, with one / several plane segments)
    variants = [dict(fuse_sweeps=0, fuse_small=0), dict(), dict(fuse_sweeps=0), dict(fuse_k=0), dict(fuse_k=0, fuse_segments=1),
                dict(fuse_k=0, fuse_segments=3), dict(fuse_classes=0), dict(fuse_classes=0, fuse_plain_shape=1),
                dict(fuse_classes=0, fuse_plain_shape=2, fuse_segments=3),
                dict(fuse_classes=0, fuse_plain=1), dict(fuse_k=0, fuse_shape=0), dict(fuse_k=0, fuse_shape=3),
                dict(fuse_k=0, fuse_shape=2, fuse_segments=2),
                dict(fuse_segments=5, fuse_nontemporal=1, fuse_classes=0), dict(rows_per_lane=1),
                dict(rows_per_lane=4, fuse_segments=2), dict(rows_per_lane=1, row_classes=0),
                dict(fuse_k=3), dict(fuse_k=4, fuse_k_segments=1), dict(fuse_k=5, fuse_k_segments=3), dict(fuse_k=4, fuse_k_shape=1),
                dict(fuse_k=5, fuse_k_shape=1, fuse_k_segments=2), dict(fuse_k=4, fuse_k_shape=2), dict(fuse_k=3, fuse_k_shape=2, fuse_k_segments=4),
                dict(fuse_k=4, fuse_k_dpp=0), dict(fuse_k=3, fuse_k_dpp=0, fuse_k_segments=2),
                dict(fuse_k=3, fuse_k_shape=3), dict(fuse_k=4, fuse_k_shape=3, fuse_k_segments=2), dict(fuse_k=5, fuse_k_shape=3),
                dict(fuse_k=3, fuse_k_shape=4, fuse_k_segments=3), dict(fuse_k=4, fuse_k_shape=4), dict(fuse_k=4, fuse_k_shape=5),
                dict(fuse_k=3, fuse_k_shape=1, fuse_k_pf=2), dict(fuse_k=3, fuse_k_shape=2, fuse_k_pf=2, fuse_k_segments=2),
                dict(fuse_k=3, fuse_k_shape=6), dict(fuse_k=4, fuse_k_shape=7, fuse_k_segments=2),
                # (the tiles of a last, nearly empty round in shorter segments: planned for 4 / 5 / 7 resident workgroups so
                #  that these few-tile grids have such a tail; 0: every tile alike)
                dict(fuse_k=5, fuse_k_tail=4), dict(fuse_k=4, fuse_k_tail=5, fuse_k_shape=1), dict(fuse_k=3, fuse_k_tail=7),
                dict(fuse_k=5, fuse_k_tail=0),
                # (the block pass of mg_jacobiblk.hip.h on every level, in all its shapes: K sweeps per launch, planes per block)
                dict(fuse_block=2), dict(fuse_block=2, fuse_block_k=2, fuse_block_ez=11), dict(fuse_block=2, fuse_block_k=3, fuse_block_ez=11),
                dict(fuse_block=2, fuse_block_k=4, fuse_block_ez=11), dict(fuse_block=2, fuse_block_k=2, fuse_block_ez=19),
                dict(fuse_block=2, fuse_block_k=3, fuse_block_ez=19), dict(fuse_block=2, fuse_block_k=4, fuse_block_ez=19)]
    for kw in variants:
        tune = {k: v for k, v in kw.items() if k.startswith("fuse_")}
        make = {k: v for k, v in kw.items() if not k.startswith("fuse_")}
        classes = make.get("row_classes", 1)
        with DeviceHierarchy.synthetic(3, lo, hi, c=c, mu1=2, mu2=2, **make) as dev:
            for key in ("fuse_min_rows", "fuse_k_min_rows", "fuse_k4_min_rows", "fuse_k5_min_rows"):       # small grids: still the march kernels
                dev.set_tuning(key, 0)
            for k, v in tune.items():
                dev.set_tuning(k, v)
            for level in range(lo + 1, hi + 1):
                info = dev.level_info(level)
                assert info["symmetric_diagonals"] == 4
                # the generated Poisson rows: interior stencil, its variants next to the boundary, identity rows
                assert (2 <= info["row_classes"] <= 64) if classes else info["row_classes"] == 0
                n = info["n_global"]
                if (level, "v") not in want:
                    want[level, "v"] = rng.standard_normal(n)
                    want[level, "f"] = rng.standard_normal(n)
                for nw in (2, 3, 4, 5, 6, 9):
                    dev.set_vector(level, "v", want[level, "v"])
                    dev.set_vector(level, "f", want[level, "f"])
                    dev.smooth(level, nw)
                    got = dev.get_vector(level, "v")
                    if kw == variants[0]:
                        want[level, nw] = got
                    else:
                        assert np.array_equal(got, want[level, nw]), (kw, level, nw)
            # whole cycles through the graph cache with an odd number of buffer swaps per smoother call
            dev.set_params(3, 5, 2.0 / 3.0)
            dev.zero_vector(hi, "v")
            res = dev.vcycle(hi, 3, residuals=True)
            if kw == variants[0]:
                want["res"] = res
            else:
                assert np.all(np.abs(res - want["res"]) <= 1e-13 * want["res"]), kw


def test_row_classes_fall_back_when_there_are_too_many_distinct_rows():
    """Symmetric 7-point matrices with many distinct rows (the Poisson matrix scaled symmetrically by random powers of
    two, which keeps it symmetric bit for bit) get no class dictionary: the two-sweep pass reads the rows themselves.
    With only a few rows rescaled the dictionary grows by a few classes.  Same results as single sweeps either way."""
    import scipy.sparse as sps
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    rng = np.random.default_rng(11)
    bag = poisson.make_hierarchy(3, 1, 3, c=5, mu1=2, mu2=2, seed=None)         # 41^3 unknowns on the finest level
    A = bag.A_sp_dict[3][0].tocsr()
    n = A.shape[0]
    v_in = rng.standard_normal(n)
    base_classes = None
    for case in ("plain", "few", "many"):
        e = np.zeros(n)
        if case == "few":
            e[rng.choice(n, 5, replace=False)] = 1.0
        elif case == "many":
            e = rng.integers(0, 8, n).astype(np.float64)
        d = sps.diags(2.0 ** e)
        B = (d @ A @ d).tocsr()
        outs = []
        for fuse in (0, 1):
            with DeviceHierarchy(3, 1, 3, c=5) as dev:
                dev.set_tuning("fuse_min_rows", 0)
                dev.set_tuning("fuse_sweeps", fuse)
                for l in (1, 2):
                    dev.set_level(l, bag.A_sp_dict[l][0], bag.levels[l].grid_index)
                dev.set_level(3, B, bag.levels[3].grid_index)
                dev.set_params(4, 4, 2.0 / 3.0)
                info = dev.level_info(3)
                assert info["symmetric_diagonals"] == 4, info
                if case == "plain":
                    base_classes = info["row_classes"]
                    assert 2 <= base_classes <= 64
                elif case == "few":
                    assert base_classes < info["row_classes"] <= base_classes + 80
                else:
                    assert info["row_classes"] == 0
                dev.set_vector(3, "v", v_in)
                dev.set_vector(3, "f", bag.b_dict[3])
                dev.smooth(3, 5)
                outs.append(dev.get_vector(3, "v"))
        assert np.array_equal(outs[0], outs[1]), case


@pytest.mark.parametrize("case", ["diagonal", "scaled", "patch10", "patch16"])
def test_escape_rows_keep_the_class_path(case):
    """A uniform-mesh matrix with some odd rows has more than 255 distinct rows but is still mostly copies of a few: the
    254 most frequent rows keep their classes, the others ("escape rows") are fetched from the stored matrix by the K-sweep
    march (mg_jacobik3d.hip.h, JK3_POOL), and the level stays on the class path -- same arithmetic as single sweeps on the
    stored rows, bit for bit, and equal to the oracle's Jacobi relaxation on the same matrix.
      diagonal  1 % of the rows, picked at random, with their diagonal entry perturbed
      scaled    D A D with 1 % of D's entries random powers of two (seven rows change per entry; symmetric bit for bit)
      patch10   a 10 x 10 x 10 block of perturbed rows: one tile meets 100 of them per plane, which fits the march's pool
                (1024 rows, 512 with five sweeps per pass) for four sweeps per pass (six planes) but not for five (seven)
      patch16   16 x 16 x 16: too many for any pass -- the level goes without classes, as before"""
    import scipy.sparse as sps
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    from oracle.mg_oracle import get_jacobi_matrices, jacobi_relaxation
    rng = np.random.default_rng(5)
    bag = poisson.make_hierarchy(3, 1, 3, c=8, mu1=2, mu2=2, seed=None)         # 65^3 unknowns on the finest level
    A = bag.A_sp_dict[3][0].tocsr().copy()
    n = A.shape[0]
    odd = np.zeros(n, dtype=bool)
    if case in ("diagonal", "scaled"):
        odd[rng.choice(n, n // 100, replace=False)] = True
    else:
        w = 10 if case == "patch10" else 16
        g = np.zeros((65, 65, 65), dtype=bool)
        g[20:20 + w, 30:30 + w, 25:25 + w] = True
        odd = g.ravel()[np.asarray(bag.levels[3].grid_index)]
    if case == "scaled":
        d = np.where(odd, 2.0 ** rng.integers(-20, 21, n), 1.0)
        B = (sps.diags(d) @ A @ sps.diags(d)).tocsr()
    else:
        B = (A + sps.diags(np.where(odd, rng.uniform(0.1, 1.0, n) * A.diagonal(), 0.0))).tocsr()
    v_in = rng.standard_normal((n, 1))
    f_in = rng.standard_normal((n, 1))
    outs = {}
    for escape in (1, 0):
        with DeviceHierarchy(3, 1, 3, c=8, row_escape=escape) as dev:
            if not escape:
                dev.set_tuning("fuse_sweeps", 0)                                # single sweeps on the stored rows
            for l in (1, 2):
                dev.set_level(l, bag.A_sp_dict[l][0], bag.levels[l].grid_index)
            dev.set_level(3, B, bag.levels[3].grid_index)
            dev.set_params(4, 4, 2.0 / 3.0)
            info, st = dev.level_info(3), dev.level_storage(3)
            assert info["symmetric_diagonals"] == 4 and st["distinct_rows"] > 255, (info, st)
            if escape and case != "patch16":
                assert info["row_classes"] == 255 and 0 < st["escape_rows"] <= 0.25 * n, (info, st)
                if case != "scaled":
                    # (the odd rows -- but for any that were picked for the dictionary's spare classes -- and the few regular rows
                    #  that the dictionary's sample met only once: rows next to a corner of the grid)
                    assert int(odd.sum()) - 254 <= st["escape_rows"] <= int(odd.sum()) + 64
                assert dev.time_kernel("jacobik3", 3, 1) > 0.0                  # the march runs on this level
            else:
                assert info["row_classes"] == 0 and st["escape_rows"] == 0, (info, st)
            for nw in (3, 4, 5, 7, 12):
                dev.set_vector(3, "v", v_in)
                dev.set_vector(3, "f", f_in)
                dev.smooth(3, nw)
                outs[escape, nw] = dev.get_vector(3, "v")
            dev.set_vector(3, "f", bag.b_dict[3])
            dev.zero_vector(3, "v")
            outs[escape, "res"] = dev.vcycle(3, 3, residuals=True)
    for nw in (3, 4, 5, 7, 12):
        assert np.array_equal(outs[1, nw], outs[0, nw]), (case, nw)
    assert np.all(np.abs(outs[1, "res"] - outs[0, "res"]) <= 1e-13 * outs[0, "res"])
    want = jacobi_relaxation(get_jacobi_matrices((B, 3)), v_in, f_in, 5, 2.0 / 3.0)
    assert rel_l2(outs[1, 5], want) <= 1e-13


@pytest.mark.parametrize("c,lo,hi", [(8, 1, 4), (5, 1, 5), (3, 2, 6)])
def test_k_sweep_kernel_on_2d_levels_is_bit_identical_to_single_sweeps(c, lo, hi):
    """mg_jacobi2.hip.h, sdia_jacobik2d: up to five Jacobi sweeps per launch on 2-D levels (a region of the grid in LDS and
    registers, the exact part shrinks by one ring per sweep) must reproduce single sweeps bit for bit -- for sweep counts that
    split into different launch sizes, grids that are not a multiple of the tile, and whole V-cycles."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    rng = np.random.default_rng(c)
    want = {}
    # (fuse_2d_lines: regions of 64 / 32 / 16 lines -- eight, four, two cells per thread; 0, the default: chosen per level)
    for kw in (dict(fuse_2d=0, fuse_small=0), dict(), dict(fuse_small=0), dict(fuse_small=0, fuse_2d_k=2), dict(fuse_2d_k=3),
               dict(fuse_small=0, fuse_2d_k=4), dict(rows_per_lane=1), dict(fuse_small=0, fuse_2d_lines=64),
               dict(fuse_small=0, fuse_2d_lines=32, fuse_2d_k=4), dict(fuse_small=0, fuse_2d_lines=16),
               dict(fuse_small=0, fuse_2d_lines=16, fuse_2d_k=3),
               # (the one-launch smoother also on the levels the 2-D kernel takes by default: more than 2048 rows)
               dict(fuse_small_2d_rows=16384)):
        tune = {k: v for k, v in kw.items() if k.startswith("fuse_")}
        make = {k: v for k, v in kw.items() if not k.startswith("fuse_")}
        with DeviceHierarchy.synthetic(2, lo, hi, c=c, mu1=2, mu2=2, **make) as dev:
            for k, v in tune.items():
                dev.set_tuning(k, v)
            for level in range(lo + 1, hi + 1):
                info = dev.level_info(level)
                assert info["symmetric_diagonals"] == 3 and info["row_classes"] > 0, info
                n = info["n_global"]
                if (level, "v") not in want:
                    want[level, "v"] = rng.standard_normal(n)
                    want[level, "f"] = rng.standard_normal(n)
                for nw in (2, 3, 5, 6, 7, 11):
                    dev.set_vector(level, "v", want[level, "v"])
                    dev.set_vector(level, "f", want[level, "f"])
                    dev.smooth(level, nw)
                    got = dev.get_vector(level, "v")
                    if not kw:
                        assert dev.time_kernel("jacobik", level, 1) > 0.0
                    if kw == dict(fuse_2d=0, fuse_small=0):
                        want[level, nw] = got
                    else:
                        assert np.array_equal(got, want[level, nw]), (kw, level, nw)
            dev.set_params(50, 49, 2.0 / 3.0)
            dev.zero_vector(hi, "v")
            res = dev.vcycle(hi, 3, residuals=True)
            if kw == dict(fuse_2d=0, fuse_small=0):
                want["res"] = res
            else:
                assert np.all(np.abs(res - want["res"]) <= 1e-13 * want["res"]), kw


def test_storage_ulps_brings_round_off_noisy_assemblies_to_the_compact_formats():
    """An assembly whose rows differ by round-off (h not a power of two, varying summation order) is neither bit-for-bit
    symmetric nor repetitive: it keeps the offset-coded format.  With `storage_ulps` = k entries that agree within ~k
    ulps count as equal in the symmetry test and the row dictionary, the level gets symmetric diagonals + row classes
    (and with them the paired pass), and the results agree with the exact-storage ones to round-off -- by construction
    not bit for bit: the knob perturbs the matrix by up to k ulps per entry and is off by default."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    rng = np.random.default_rng(21)
    bag = poisson.make_hierarchy(3, 1, 3, c=5, mu1=2, mu2=2, seed=None)         # 41^3 unknowns on the finest level
    A = bag.A_sp_dict[3][0].tocsr().copy()
    ulp = 2.0 ** -52
    A.data = A.data * (1.0 + ulp * rng.integers(-1, 2, A.data.size))            # every entry off by -1, 0 or +1 ulp
    n = A.shape[0]
    v_in = rng.standard_normal(n)
    outs = {}
    for ulps in (0, 8, "auto"):
        # (0: exact storage only, the automatic second try off; "auto": the defaults -- a level whose exact symmetry test or
        #  row dictionary fails by at most 4 ulps is taken with that tolerance, and says so in mg_level_storage)
        kw = dict(storage_auto=0) if ulps == 0 else dict(storage_ulps=ulps) if ulps == 8 else {}
        with DeviceHierarchy(3, 1, 3, c=5, **kw) as dev:
            dev.set_tuning("fuse_min_rows", 0)
            for l in (1, 2):
                dev.set_level(l, bag.A_sp_dict[l][0], bag.levels[l].grid_index)
            dev.set_level(3, A, bag.levels[3].grid_index)
            dev.set_params(4, 4, 2.0 / 3.0)
            info = dev.level_info(3)
            st = dev.level_storage(3)
            if ulps == 0:
                assert info["symmetric_diagonals"] == 0 and info["offset_codes"] == 7 and info["row_classes"] == 0, info
                # ... and why: the first row with a pair that is not symmetric bit for bit, by how many ulps at most
                assert st["symmetric"] == 0 and st["first_asymmetric_row"] >= 0 and 1 <= st["max_pair_ulps"] <= 4 and st["ulps_used"] == 0, st
            else:
                assert info["symmetric_diagonals"] == 4 and 2 <= info["row_classes"] <= 255, info
                assert dev.time_kernel("jacobi2", 3, 1) > 0.0
                assert st["symmetric"] == 2 and st["ulps_used"] == (8 if ulps == 8 else 4) and 2 <= st["distinct_rows"] <= 255, st
            # (an exactly symmetric, repetitive level: nothing identified -- unless "storage_ulps" asks for the tolerance everywhere)
            assert dev.level_storage(2) == dict(symmetric=1, first_asymmetric_row=-1, max_pair_ulps=0, ulps_used=8 if ulps == 8 else 0,
                                                distinct_rows=dev.level_info(2)["row_classes"] - 1, escape_rows=0)
            dev.set_vector(3, "v", v_in)
            dev.set_vector(3, "f", bag.b_dict[3])
            dev.smooth(3, 6)
            sm = dev.get_vector(3, "v")
            dev.zero_vector(3, "v")
            res = dev.vcycle(3, 3, residuals=True)
            outs[ulps] = (sm, res, dev.get_vector(3, "v"))
    for k in (8, "auto"):
        assert rel_l2(outs[k][0], outs[0][0]) <= 1e-13
        assert np.all(np.abs(outs[k][1] - outs[0][1]) <= 1e-11 * outs[0][1])
        assert rel_l2(outs[k][2], outs[0][2]) <= 1e-12


def test_time_kernel_reports_where_the_two_sweep_pass_is_not_used():
    """`mg_time_kernel("jacobi2")` is how bench.py finds out whether the smoother pairs sweeps on a level: an error on
    levels where it does not (2-D, too small), a duration where it does."""
    from multigrid_dolfinx_amd._capi import MgError
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    with DeviceHierarchy.synthetic(2, 1, 3, c=8, mu1=2, mu2=2) as dev:
        with pytest.raises(MgError):
            dev.time_kernel("jacobi2", 3, 1)
    with DeviceHierarchy.synthetic(3, 2, 4, c=8, mu1=2, mu2=2) as dev:          # 129^3 < fuse_min_rows
        with pytest.raises(MgError):
            dev.time_kernel("jacobi2", 4, 1)
        assert dev.time_kernel("jacobi2!", 4, 1) > 0.0
        dev.set_tuning("fuse_min_rows", 0)
        assert dev.time_kernel("jacobi2", 4, 1) > 0.0


def test_3d_cycles_converge_to_the_direct_solution():
    """The 3-D path has no reference implementation (its oracle is the dimension-consistent extension of the 2-D
    one), so besides oracle parity it is pinned to what the cycle is FOR: V(50,50) cycles on a 4-level 3-D hierarchy
    -- with the paired class-coded smoother forced on -- converge monotonically to SciPy's direct solution of
    A u = f.  (Slowly: the reference injects the finite-element residual, SURVEY.md App. A Q1, which under-scales
    the coarse correction by 2^d; that is why it runs 50 sweeps per leg.  The rate is the algorithm's, not checked.)"""
    import scipy.sparse.linalg as spla
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    bag = poisson.make_hierarchy(3, 1, 4, c=2, mu1=50, mu2=50, seed=5)             # 5^3 ... 33^3 unknowns, permuted
    gi = {l: L.grid_index for l, L in bag.levels.items()}
    A, f = bag.A_sp_dict[4][0].tocsc(), bag.b_dict[4]
    u = spla.spsolve(A, f.ravel())
    with DeviceHierarchy.from_bag(bag, dim=3, grid_index=gi, fuse_min_rows=0) as dev:
        assert dev.time_kernel("jacobi2", 4, 1) > 0.0                               # the 33^3 level pairs its sweeps
        dev.zero_vector(4, "v")
        dev.set_vector(4, "f", f)
        res = dev.vcycle(4, 150, residuals=True)
        v = dev.get_vector(4, "v").ravel()
    assert np.all(res[1:] < res[:-1]) or res[-1] <= 1e-13 * np.linalg.norm(f)
    assert res[-1] <= 1e-10 * np.linalg.norm(f), res[-1] / np.linalg.norm(f)
    assert np.linalg.norm(v - u) <= 1e-9 * np.linalg.norm(u)


@pytest.mark.parametrize("c,lo,hi", [(8, 2, 4), (5, 1, 4)])
def test_one_sweep_kernels_through_row_classes_are_bit_identical(c, lo, hi):
    """`class_sweeps`: residual, single Jacobi sweeps, Gauss-Seidel colours and the SpMV + dot of the PCG / quadratic
    form read one class byte per row instead of the row where a level has row classes -- as persistent slice kernels
    and, on whole 3-D levels, as a plane march with x in LDS (`march_sweeps`, both launch shapes).  Same entries in the
    same order: bit-identical to the plain symmetric-diagonal kernels, for every rows-per-lane setting and for grids that
    are not a multiple of the tile."""
    from multigrid_dolfinx_amd.hierarchy import DeviceHierarchy
    rng = np.random.default_rng(3)
    for R in (1, 2, 4):
        outs = []
        variants = [dict(class_sweeps=0), dict(march_sweeps=0), dict(march_min_rows=0), dict(march_min_rows=0, march_shape=1)]
        for kw in variants if R == 2 else variants[:3]:
            with DeviceHierarchy.synthetic(3, lo, hi, c=c, mu1=3, mu2=3, rows_per_lane=R, fuse_sweeps=0,
                                           coarse_direct=0, **kw) as dev:                   # PCG coarsest solve: SpMV + dot
                n = dev.level_info(hi)["n_global"]
                assert dev.level_info(hi)["row_classes"] > 0
                if not outs:
                    v_in, f_in = rng.standard_normal(n), rng.standard_normal(n)
                got = []
                dev.set_vector(hi, "v", v_in)
                dev.set_vector(hi, "f", f_in)
                dev.smooth(hi, 3)
                got.append(dev.get_vector(hi, "v"))
                dev.residual(hi)
                got.append(dev.get_vector(hi, "r"))
                got.append(np.array([dev.quadratic_form(hi, "v")]))
                dev.zero_vector(hi, "v")
                got.append(np.asarray(dev.vcycle(hi, 2, residuals=True)))
                got.append(dev.get_vector(hi, "v"))
                for sm in ("rbgs", "mcgs"):
                    dev.set_params(2, 2, 1.0, smoother=sm)
                    dev.zero_vector(hi, "v")
                    dev.vcycle(hi, 1)
                    got.append(dev.get_vector(hi, "v"))
                outs.append(got)
        for other in outs[1:]:
            for a, b in zip(outs[0], other):
                assert np.array_equal(a, b), R
