"""Shared helpers for the parity tests (test infrastructure)."""
import os
import types

import numpy as np
import scipy.sparse as sp

from multigrid_dolfinx_amd import poisson

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (d if d > 0 else 1.0))


def bag_from_fixture(g):
    """Rebuild the Var_initializer-shaped bag from a self-contained fixture (c1_*)."""
    lo, hi = (int(x) for x in g["meta_levels"])
    mu0, mu1, mu2 = (int(x) for x in g["meta_mu"])
    bag = types.SimpleNamespace(
        mesh_dof_list_dict={}, element_size={}, coarsest_level_elements_per_dim=int(g["meta_c"]),
        coarsest_level=lo, finest_level=hi, A_sp_dict={}, A_jacobi_sp_dict={}, b_dict={},
        mu0=mu0, mu1=mu1, mu2=mu2, omega=float(g["meta_omega"]),
        residual_per_V_cycle_finest=[], error_per_V_cycle_finest=[], u_exact_fine=None, V_fine_dolfx=None)
    grid_index, coords = {}, {}
    for l in range(lo, hi + 1):
        n = g[f"b{l}"].shape[0]
        A = sp.csr_matrix((g[f"A{l}_data"], g[f"A{l}_indices"], g[f"A{l}_indptr"]), shape=(n, n))
        bag.A_sp_dict[l] = (A, l)
        bag.b_dict[l] = g[f"b{l}"]
        bag.element_size[l] = 1.0 / (bag.coarsest_level_elements_per_dim * 2 ** l)
        grid_index[l] = g[f"grid_index{l}"]
        coords[l] = g[f"coords{l}"]
    return bag, grid_index, coords


def hierarchy_for(g, dim=2):
    """Regenerate the (deterministic) inputs of a cycles-only fixture."""
    lo, hi = (int(x) for x in g["meta_levels"])
    mu0, mu1, mu2 = (int(x) for x in g["meta_mu"])
    seed = int(g["meta_seed"])
    return poisson.make_hierarchy(dim, lo, hi, c=int(g["meta_c"]), mu0=mu0, mu1=mu1, mu2=mu2,
                                  omega=float(g["meta_omega"]), seed=None if seed < 0 else seed)
