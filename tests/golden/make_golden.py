"""Generate golden vectors by running the reference's own `multigrid.py`.

Runs ONLY in the build container (it needs `/root/reference`); the fixtures it
writes (`tests/golden/*.npz`) are data -- inputs and the reference's outputs -- and are
what travels.  Nothing from the reference's source is copied.

How the reference is loaded: `multigrid.py:2-3` import `dolfinx` and `ufl`
unconditionally, but use them only inside `res_calculator` / `err_calculator`
(`multigrid.py:203-218`), which are off the V-cycle path and never called here.
Neither package is installable offline, so two EMPTY module objects are registered
under those names before the unmodified file is imported from where it lies.  They
implement nothing; every number below is computed by the reference's code on
NumPy/SciPy.  `FullMultiGrid` (non-test) is not runnable this way (its stop test needs
dolfinx) and is therefore not captured.

Inputs are synthetic hierarchies in dolfinx hand-off conventions
(`multigrid_dolfinx_amd/poisson.py`), because dolfinx cannot assemble them here.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REFERENCE = "/root/reference/multigrid.py"

from multigrid_dolfinx_amd import poisson  # noqa: E402


def load_reference():
    for name in ("dolfinx", "ufl"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    spec = importlib.util.spec_from_file_location("reference_multigrid", REFERENCE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def digest(h):
    m = hashlib.sha256()
    for l in sorted(h.levels):
        A = h.levels[l].A
        for arr in (A.indptr, A.indices, A.data, h.levels[l].b, h.levels[l].grid_index):
            m.update(np.ascontiguousarray(arr).tobytes())
    return m.hexdigest()


def l2(x):
    return float(np.sqrt(np.sum(np.asarray(x) ** 2)))


def prepare(ref, dim, lo, hi, seed, mu0, mu1, mu2, omega=2.0 / 3.0):
    h = poisson.make_hierarchy(dim, lo, hi, c=8, mu0=mu0, mu1=mu1, mu2=mu2, omega=omega,
                               seed=seed, with_dicts=True)
    for l, a in h.A_sp_dict.items():
        h.A_jacobi_sp_dict[l] = ref.getJacobiMatrices(a)
    ref.initialize_problem(h)
    return h


def capture_full(ref, name, seed):
    """C1: every function on the path, inputs stored so the fixture is self-contained."""
    h = prepare(ref, 2, 1, 3, seed, mu0=2, mu1=50, mu2=50)
    out = {"meta_levels": np.array([1, 3]), "meta_c": np.array(8), "meta_mu": np.array([2, 50, 50]),
           "meta_omega": np.array(2.0 / 3.0), "meta_dim": np.array(2), "inputs_sha256": np.array(digest(h))}
    rng = np.random.default_rng(1234)
    for l in range(1, 4):
        L = h.levels[l]
        out[f"A{l}_indptr"], out[f"A{l}_indices"], out[f"A{l}_data"] = L.A.indptr, L.A.indices, L.A.data
        out[f"b{l}"], out[f"coords{l}"], out[f"grid_index{l}"] = L.b, L.coords, L.grid_index
        R, Dinv, lev = h.A_jacobi_sp_dict[l]
        assert lev == l
        out[f"J{l}_indptr"], out[f"J{l}_indices"], out[f"J{l}_data"] = R.indptr, R.indices, R.data
        out[f"J{l}_sorted"] = np.array(bool(R.has_sorted_indices))
        out[f"Dinv{l}"] = Dinv.diagonal()
    A3 = h.A_jacobi_sp_dict[3]
    v0 = rng.standard_normal((h.levels[3].n, 1))
    out["jac_v0"] = v0
    v0_copy = v0.copy()
    out["jac_nw1"] = ref.jacobiRelaxation(A3, v0, h.b_dict[3], 1)
    out["jac_nw50"] = ref.jacobiRelaxation(A3, v0, h.b_dict[3], 50)
    assert np.array_equal(v0, v0_copy)          # the reference does not mutate its input
    for lc in (1, 2):
        lf = lc + 1
        xc = rng.standard_normal((h.levels[lc].n, 1))
        xf = rng.standard_normal((h.levels[lf].n, 1))
        out[f"xfer_xc{lc}"], out[f"xfer_xf{lf}"] = xc, xf
        out[f"interp_{lc}to{lf}"] = ref.Interpolation2D(
            xc, h.mesh_dof_list_dict[lc], h.mesh_dof_list_dict[lf], h.element_size[lc],
            h.element_size[lf], h.levels[lf].n)
        out[f"inject_{lf}to{lc}"] = ref.Restriction2D_direct(
            xf, h.mesh_dof_list_dict[lc], h.mesh_dof_list_dict[lf], h.levels[lc].n)
        out[f"fullw_{lf}to{lc}"] = ref.Restriction2D(
            xf, h.mesh_dof_list_dict[lc], h.mesh_dof_list_dict[lf], h.element_size[lc],
            h.element_size[lf], h.levels[lc].n)
    # V-cycles from a zero initial guess
    f = h.b_dict[3]
    v = np.zeros_like(f)
    res = []
    for k in range(1, 4):
        v = ref.V_cycle_scheme(A3, v, f)
        out[f"vcycle_iter{k}"] = v
        res.append(l2(f - h.A_sp_dict[3][0].dot(v)))
    out["vcycle_res_l2"] = np.array(res)
    t = ref.V_cycle_scheme(A3, np.zeros_like(f), f, True)
    for key, arr in zip(("v_h", "f_2h", "v_2h", "err_h"), t):
        out[f"vcycle_test_{key}"] = arr
    # a V-cycle entered on the middle level
    out["vcycle_mid_level2"] = ref.V_cycle_scheme(h.A_jacobi_sp_dict[2], np.zeros_like(h.b_dict[2]), h.b_dict[2])
    t = ref.FullMultiGrid_test(A3, f, True)
    for key, arr in zip(("v_h", "f_2h", "v_2h", "err_h"), t):
        out[f"fmg_test_{key}"] = arr
    # Reference quirk (Q9): with test=False the finest level still unpacks four values
    # from V_cycle_scheme's single (n, 1) array (`multigrid.py:331-333`) -> ValueError.
    try:
        ref.FullMultiGrid_test(A3, f, False)
        out["fmg_test_false_raises"] = np.array("")
    except ValueError as e:
        out["fmg_test_false_raises"] = np.array("ValueError: " + str(e))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written;", len(out), "arrays; residuals", res)


def capture_cycles(ref, name, lo, hi, seed, mu, ncycles, stride=1, fmg=False):
    """Larger hierarchies: V-cycle iterates (optionally a strided sample) + l2 residuals."""
    h = prepare(ref, 2, lo, hi, seed, mu0=2, mu1=mu, mu2=mu)
    out = {"meta_levels": np.array([lo, hi]), "meta_c": np.array(8), "meta_mu": np.array([2, mu, mu]),
           "meta_omega": np.array(2.0 / 3.0), "meta_dim": np.array(2), "meta_stride": np.array(stride),
           "meta_seed": np.array(-1 if seed is None else seed), "inputs_sha256": np.array(digest(h))}
    A = h.A_jacobi_sp_dict[hi]
    f = h.b_dict[hi]
    v = np.zeros_like(f)
    res, nrm = [], []
    for k in range(1, ncycles + 1):
        v = ref.V_cycle_scheme(A, v, f)
        out[f"vcycle_iter{k}"] = v[::stride].copy()
        res.append(l2(f - h.A_sp_dict[hi][0].dot(v)))
        nrm.append(l2(v))
    out["vcycle_res_l2"], out["vcycle_l2"] = np.array(res), np.array(nrm)
    if fmg:
        t = ref.FullMultiGrid_test(A, f, True)
        for key, arr in zip(("v_h", "f_2h", "v_2h", "err_h"), t):
            out[f"fmg_test_{key}"] = arr[::stride].copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "written; residuals", res)


def main():
    ref = load_reference()
    capture_full(ref, "c1_lex", None)
    capture_full(ref, "c1_perm", 0)
    capture_cycles(ref, "n128_mu2_perm", 2, 4, 0, mu=2, ncycles=2, fmg=True)
    capture_cycles(ref, "n256_mu50_lex", 3, 5, None, mu=50, ncycles=1)
    capture_cycles(ref, "n512_mu2_lex", 3, 6, None, mu=2, ncycles=2, stride=37)


if __name__ == "__main__":
    main()
