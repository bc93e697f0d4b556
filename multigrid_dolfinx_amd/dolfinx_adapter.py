"""Adapter from live dolfinx objects to the inputs of this package.

The reference's driver builds its level inputs like this (`Multigrid_prototype.py:62-118`):

    V_i = dolfinx.FunctionSpace(mesh_i, ("CG", 1));  coords = V_i.tabulate_dof_coordinates()      # :67-68
    A_i = dolfinx.fem.assemble_matrix(a_i, bcs=[bc_i]); A_i.assemble()                             # :92-93
    ai, aj, av = A_i.getValuesCSR();  A_sp_i = scipy.sparse.csr_matrix((av, aj, ai))               # :95-96
    b_dict[i] = np.array(b_i.array).reshape((num_elems_i + 1) ** 2, 1)                             # :110

The functions below take those same objects (duck-typed: anything with `getValuesCSR()` /
`tabulate_dof_coordinates()` / `.array`) and produce the CSR, the right-hand side and the integer grid index
that replace the coordinate dictionaries.  dolfinx is not installable in the build container, so this module
is exercised with stand-in objects exposing exactly those methods (tests/test_dolfinx_adapter.py); it has not
been run against a real dolfinx build.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict, Optional, Sequence

import numpy as np
import scipy.sparse as sp

from .poisson import grid_index_from_coords

__all__ = ["csr_from_petsc", "coordinates_of", "level_from_dolfinx", "bag_from_dolfinx", "mass_matrix_of"]


def csr_from_petsc(A) -> sp.csr_matrix:
    """PETSc `Mat` (or anything with `getValuesCSR()`) -> SciPy CSR exactly as `Multigrid_prototype.py:95-96`
    builds it; an already-SciPy matrix is passed through (converted to CSR)."""
    if sp.issparse(A):
        return A.tocsr()
    ai, aj, av = A.getValuesCSR()
    av = np.asarray(av)
    if np.iscomplexobj(av):
        raise TypeError("complex-scalar PETSc builds are not supported (real fp64 only)")
    return sp.csr_matrix((np.asarray(av, dtype=np.float64), np.asarray(aj, dtype=np.int32),
                          np.asarray(ai, dtype=np.int32)))


def coordinates_of(V, n: Optional[int] = None) -> np.ndarray:
    """DoF coordinates `(n, 3)` of a function space (`tabulate_dof_coordinates()`, old and current dolfinx
    API), restricted to the locally owned DoFs like the reference's loop bound
    (`size_local * index_map_bs`, `Multigrid_prototype.py:70`); a plain array is passed through."""
    if isinstance(V, np.ndarray):
        coords = V
    else:
        coords = np.asarray(V.tabulate_dof_coordinates())
        if n is None:
            try:
                im = V.dofmap.index_map
                n = int(im.size_local) * int(getattr(V.dofmap, "index_map_bs", 1))
            except AttributeError:
                n = None
    coords = np.asarray(coords, dtype=np.float64)
    if coords.ndim != 2:
        raise ValueError("DoF coordinates must be a 2-D array")
    if coords.shape[1] == 2:                       # some versions return (n, gdim)
        coords = np.hstack([coords, np.zeros((coords.shape[0], 1))])
    return coords[:n] if n is not None else coords


def level_from_dolfinx(A, V, b=None, elements_per_dim: Optional[int] = None, dim: int = 2):
    """One level: returns `(csr, grid_index, rhs, coords)`.

    `elements_per_dim` defaults to `round(n ** (1/dim)) - 1` (the reference's unit-square meshes have
    `(N+1)**2` DoFs, `Multigrid_prototype.py:110`)."""
    csr = csr_from_petsc(A)
    n = csr.shape[0]
    coords = coordinates_of(V, n)
    if coords.shape[0] != n:
        raise ValueError(f"{coords.shape[0]} DoF coordinates for a {n}-row matrix (parallel dolfinx runs are "
                         "not supported: the reference is serial-only)")
    N = elements_per_dim if elements_per_dim is not None else int(round(n ** (1.0 / dim))) - 1
    if (N + 1) ** dim != n:
        raise ValueError("the DoFs do not form an (N+1)^dim grid")
    gi = grid_index_from_coords(coords, N, dim)
    rhs = None
    if b is not None:
        rhs = np.array(getattr(b, "array", b), dtype=np.float64).reshape(n, 1)
    return csr, gi, rhs, coords


def bag_from_dolfinx(levels: Dict[int, Sequence], coarsest_level_elements_per_dim: int, mu0: int = 2,
                     mu1: int = 50, mu2: int = 50, omega: float = 2.0 / 3.0, dim: int = 2):
    """`{level: (A, V, b)}` -> `(bag, grid_index)` where `bag` has the sixteen `Var_initializer` fields
    (`Multigrid_prototype.py:15-32`) and can go straight to `initialize_problem` after
    `configure(grid_index=grid_index, dim=dim)`; no coordinate dictionaries are built."""
    keys = sorted(levels)
    bag = SimpleNamespace(
        mesh_dof_list_dict={}, element_size={}, coarsest_level_elements_per_dim=coarsest_level_elements_per_dim,
        coarsest_level=keys[0], finest_level=keys[-1], A_sp_dict={}, A_jacobi_sp_dict={}, b_dict={},
        mu0=mu0, mu1=mu1, mu2=mu2, omega=omega, residual_per_V_cycle_finest=[], error_per_V_cycle_finest=[],
        u_exact_fine=None, V_fine_dolfx=None)
    grid_index = {}
    for l in keys:
        A, V, b = levels[l]
        N = coarsest_level_elements_per_dim * 2 ** l
        csr, gi, rhs, _ = level_from_dolfinx(A, V, b, elements_per_dim=N, dim=dim)
        bag.A_sp_dict[l] = (csr, l)
        bag.A_jacobi_sp_dict[l] = (None, None, l)        # only the level key is used by the device path
        bag.b_dict[l] = rhs
        bag.element_size[l] = 1.0 / N
        grid_index[l] = gi
    return bag, grid_index


def mass_matrix_of(V) -> sp.csr_matrix:
    """P1 mass matrix `M_ij = int phi_i phi_j dx` of a dolfinx function space, so that `sqrt(r^T M r)` is the
    reference's `res_calculator` norm `sqrt(assemble_scalar(r_h * r_h * dx))` (`multigrid.py:203-208`) for the
    finite-element function with nodal values r.  Needs an importable dolfinx + ufl (2021-era or current API); a
    SciPy matrix is passed through.  Never run against a real dolfinx build here (not installable offline)."""
    if sp.issparse(V):
        return V.tocsr()
    try:
        import dolfinx
        import ufl
    except ImportError as exc:
        raise TypeError("V_fine_dolfx is neither a SciPy mass matrix nor usable without dolfinx/ufl: pass the P1 mass "
                        "matrix of the finest mesh (scipy.sparse) as V_fine_dolfx, or configure(norm='l2')") from exc
    u, v = ufl.TrialFunction(V), ufl.TestFunction(V)
    form = u * v * ufl.dx
    fem = dolfinx.fem
    if hasattr(fem, "form"):                          # current API
        A = fem.petsc.assemble_matrix(fem.form(form))
    else:                                             # the reference's API (Multigrid_prototype.py:92)
        A = fem.assemble_matrix(form)
    A.assemble()
    return csr_from_petsc(A)
