"""Drop-in for the reference's `multigrid.py`: same function names, argument order and
return shapes, with the V-cycle body running as HIP kernels on an MI355X.

    from multigrid_dolfinx_amd.multigrid import (getJacobiMatrices, initialize_problem,
                                                 FullMultiGrid, FullMultiGrid_test, V_cycle_scheme)

replaces `from multigrid import ...` in `Multigrid_prototype.py:8`.  The sixteen module
globals of the reference (`multigrid.py:10-25`) are kept so that scripts which poke at
them keep working; the arithmetic state lives in one device-resident
`DeviceHierarchy`, built lazily from those globals after `initialize_problem`.

Differences from the reference, all documented in DESIGN.md:
  * there is no CPU path: without `libmg_hip.so` and a GPU every function raises;
  * the smoother streams the level's stiffness matrix `A_sp_dict[level]` in the
    one-matrix form `v + w D^-1 (f - A v)` (identical to the split form to ~2e-16 per
    sweep); the `A_h` argument selects the level (`A_h[2]`), as `A_sp_dict` already does
    for the residual and the coarsest solve in the reference (`multigrid.py:239`, `:244`);
  * the coarsest level is solved exactly by a block-tridiagonal LU on the device (Jacobi-PCG to
    1e-14 where the blocks would not fit) instead of SuperLU;
  * the stop test and the histories of `FullMultiGrid` use the reference's L2(Omega) norm as
    `sqrt(r^T M r)` with the P1 mass matrix M of the finest mesh given as `V_fine_dolfx` (or
    assembled from a dolfinx function space by `dolfinx_adapter.mass_matrix_of`); the Euclidean
    norm only on request (`configure(norm="l2")`, or `V_fine_dolfx = None`).
Results agree with the reference to <= 1e-10 relative l2 (tests/test_gpu_parity.py).
"""
from __future__ import annotations

import csv

import numpy as np
import scipy.sparse as sp

from .hierarchy import DeviceHierarchy, jacobi_split
from .poisson import grid_index_from_coords

# ---- the reference's module state (multigrid.py:10-25) --------------------------------------------
mesh_dof_list_dict = None
element_size = None
coarsest_level_elements_per_dim = None
coarsest_level = None
finest_level = None
A_sp_dict = None
A_jacobi_sp_dict = None
b_dict = None
mu0 = None
mu1 = None
mu2 = None
omega = None
residual_per_V_cycle_finest = None
error_per_V_cycle_finest = None
u_exact_fine = None
V_fine_dolfx = None

# ---- state of this implementation --------------------------------------------------------------------
_options = {"dim": 2, "prune_zeros": True, "device": 0, "restriction": "direct", "smoother": "jacobi",
            "grid_index": None, "norm": "auto",
            "coarse_rtol": 1e-14, "stop_tol": 1e-11, "max_cycles": 10000, "tuning": {}}
_hier = None            # DeviceHierarchy of the initialised problem
_hier_params = None     # the parameter tuple last sent to it (mg_set_params only when it changes)
_CACHE_MAX = 8


class _LRU(dict):
    """Bounded cache of device contexts / grid indices: the least recently used entry goes first and is closed."""

    def __init__(self, close=None):
        super().__init__()
        self._close = close

    def lookup(self, key):
        hit = self.get(key)
        if hit is not None:
            self[key] = self.pop(key)           # most recently used last
        return hit

    def store(self, key, value):
        self.pop(key, None)
        while len(self) >= _CACHE_MAX:
            old = self.pop(next(iter(self)))
            if self._close:
                self._close(old)
        self[key] = value
        return value

    def drop_all(self):
        while self:
            old = self.pop(next(iter(self)))
            if self._close:
                self._close(old)


_grid_cache = _LRU()                                    # id(mesh dict) -> (dict, grid_index, N)
_adhoc = _LRU(close=lambda hit: _release(hit[0]))       # stand-alone device contexts (transfers / smoother / norms)


def configure(**kw):
    """Options with no reference counterpart: `dim` (2 or 3), `prune_zeros`, `device`,
    `restriction` ('direct' = the live path, or 'full_weighting'), `smoother` ('jacobi' = the
    reference's, or 'rbgs' = red-black Gauss-Seidel with `omega` as SOR factor), `grid_index`
    ({level: lexicographic node index per DoF}, instead of coordinate dictionaries),
    `norm` ('auto': the reference's L2(Omega) norm through the mass matrix `V_fine_dolfx`, the l2 norm when that is
    None; 'l2': always the l2 norm), `coarse_rtol`, `stop_tol`, `max_cycles`, `tuning` (kernel knobs)."""
    global _hier, _hier_params
    for k, v in kw.items():
        if k not in _options:
            raise KeyError(f"unknown option {k}")
        _options[k] = v
    _release(_hier)
    _hier = _hier_params = None
    _adhoc.drop_all()
    _grid_cache.drop_all()


def _release(h):
    if h is not None:
        h.close()


def initialize_problem(obj):
    """Copy the 16-field bag into module state (`multigrid.py:28-45`); the device hierarchy is
    (re)built on the next hot-path call."""
    global mesh_dof_list_dict, element_size, coarsest_level_elements_per_dim, coarsest_level, finest_level
    global A_sp_dict, A_jacobi_sp_dict, b_dict, mu0, mu1, mu2, omega, residual_per_V_cycle_finest
    global error_per_V_cycle_finest, u_exact_fine, V_fine_dolfx, _hier, _hier_params
    mesh_dof_list_dict = obj.mesh_dof_list_dict
    element_size = obj.element_size
    coarsest_level_elements_per_dim = obj.coarsest_level_elements_per_dim
    coarsest_level = obj.coarsest_level
    finest_level = obj.finest_level
    A_sp_dict = obj.A_sp_dict
    A_jacobi_sp_dict = obj.A_jacobi_sp_dict
    b_dict = obj.b_dict
    mu0 = obj.mu0
    mu1 = obj.mu1
    mu2 = obj.mu2
    omega = obj.omega
    residual_per_V_cycle_finest = obj.residual_per_V_cycle_finest
    error_per_V_cycle_finest = obj.error_per_V_cycle_finest
    u_exact_fine = obj.u_exact_fine
    V_fine_dolfx = obj.V_fine_dolfx
    _release(_hier)
    _hier = _hier_params = None
    _adhoc.drop_all()
    _grid_cache.drop_all()


def _grid_index_of(mesh_dict, N, dim):
    """Lexicographic node index per DoF from a reference coordinate dictionary."""
    key = id(mesh_dict)
    hit = _grid_cache.lookup(key)
    if hit is not None and hit[0] is mesh_dict and hit[2] == N:
        return hit[1]
    n = (N + 1) ** dim
    coords = np.array([mesh_dict[j] for j in range(n)], dtype=np.float64)
    gi = grid_index_from_coords(coords, N, dim)
    _grid_cache.store(key, (mesh_dict, gi, N))
    return gi


def _hierarchy():
    """The device hierarchy of the initialised problem (built once)."""
    global _hier, _hier_params
    if A_sp_dict is None:
        raise RuntimeError("initialize_problem has not been called")
    if _hier is None:
        dim = _options["dim"]
        h = DeviceHierarchy(dim, coarsest_level, finest_level, c=coarsest_level_elements_per_dim,
                            device=_options["device"], **_options["tuning"])
        for level in range(coarsest_level, finest_level + 1):
            gi = None
            if _options["grid_index"] is not None:
                gi = _options["grid_index"][level]
            elif mesh_dof_list_dict:
                gi = _grid_index_of(mesh_dof_list_dict[level], h.elements(level), dim)
            h.set_level(level, A_sp_dict[level][0], gi, prune_zeros=_options["prune_zeros"])
        _hier, _hier_params = h, None
    # the parameters go to the device only when they change (the captured V-cycle graphs depend on them)
    params = (mu1, mu2, omega, _options["restriction"], _options["coarse_rtol"], _options["smoother"])
    if params != _hier_params:
        _hier.set_params(mu1, mu2, omega, restriction=_options["restriction"],
                         coarse_rtol=_options["coarse_rtol"], keep_err=True, smoother=_options["smoother"])
        _hier_params = params
    return _hier


def _column(x):
    return np.asarray(x, dtype=np.float64).reshape(-1, 1)


# ---- set-up -----------------------------------------------------------------------------------------------
def getJacobiMatrices(A):
    """`(A, level) -> (D^-1 (A - D), D^-1, level)` (`multigrid.py:48-56`), computed on the GPU."""
    R, Dinv = jacobi_split(A[0], device=_options["device"])
    return (R, Dinv, A[1])


# ---- transfers as free functions (the reference's signatures) ---------------------------------------------------
def _transfer_context(mesh_dict_coarse, mesh_dict_fine, n_coarse, n_fine):
    dim = _options["dim"]
    Nc = int(round(n_coarse ** (1.0 / dim))) - 1
    Nf = int(round(n_fine ** (1.0 / dim))) - 1
    if (Nc + 1) ** dim != n_coarse or (Nf + 1) ** dim != n_fine or Nf != 2 * Nc:
        raise ValueError("vector sizes do not describe two nested (N+1)^dim grids")
    key = ("xfer", id(mesh_dict_coarse), id(mesh_dict_fine), Nc)
    hit = _adhoc.lookup(key)
    if hit is not None and hit[1] is mesh_dict_coarse and hit[2] is mesh_dict_fine:
        return hit[0]
    h = DeviceHierarchy(dim, 0, 1, c=Nc, device=_options["device"])
    h.set_level_grid(0, _grid_index_of(mesh_dict_coarse, Nc, dim))
    h.set_level_grid(1, _grid_index_of(mesh_dict_fine, Nf, dim))
    _adhoc.store(key, (h, mesh_dict_coarse, mesh_dict_fine))
    return h


def Interpolation2D(vec_2h, mesh_dict_coarse, mesh_dict_fine, element_size_coarse, element_size_fine, vec_h_dim):
    """Q1 prolongation (`multigrid.py:59-120`); returns a new `(vec_h_dim, 1)` array."""
    vec_2h = _column(vec_2h)
    h = _transfer_context(mesh_dict_coarse, mesh_dict_fine, vec_2h.shape[0], int(vec_h_dim))
    h.set_vector(0, "v", vec_2h)
    h.prolong(1, add=False)
    return h.get_vector(1, "err")


def Restriction2D_direct(vec_h, mesh_dict_coarse, mesh_dict_fine, vec_2h_dim):
    """Injection (`multigrid.py:123-132`); returns a new `(vec_2h_dim, 1)` array."""
    vec_h = _column(vec_h)
    h = _transfer_context(mesh_dict_coarse, mesh_dict_fine, int(vec_2h_dim), vec_h.shape[0])
    h.set_vector(1, "r", vec_h)
    h.restrict(1, "direct")
    return h.get_vector(0, "f")


def Restriction2D(vec_h, mesh_dict_coarse, mesh_dict_fine, element_size_coarse, element_size_fine, vec_2h_dim):
    """Full weighting (`multigrid.py:135-198`); returns a new `(vec_2h_dim, 1)` array."""
    vec_h = _column(vec_h)
    h = _transfer_context(mesh_dict_coarse, mesh_dict_fine, int(vec_2h_dim), vec_h.shape[0])
    h.set_vector(1, "r", vec_h)
    h.restrict(1, "full_weighting")
    return h.get_vector(0, "f")


Interpolation3D = Interpolation2D               # same kernels; configure(dim=3) selects trilinear weights
Restriction3D_direct = Restriction2D_direct
Restriction3D = Restriction2D


# ---- diagnostics -------------------------------------------------------------------------------------------------------
def res_calculator(res, V_space):
    """`sqrt(int r_h^2)` in the reference (`multigrid.py:203-208`, dolfinx).  Here `V_space` may be a
    SciPy mass matrix M (returns `sqrt(r^T M r)`) or None (l2 norm)."""
    r = _column(res)
    if V_space is None:
        key = ("l2", r.shape[0])
        hit = _adhoc.lookup(key)
        if hit is None:
            h = DeviceHierarchy(_options["dim"], 0, 0, c=1, device=_options["device"])
            h.set_flat_space(r.shape[0])
            hit = _adhoc.store(key, (h, None, None))
        hit[0].set_vector(0, "v", r)
        return hit[0].norm2(0, "v")
    if not sp.issparse(V_space):
        from .dolfinx_adapter import mass_matrix_of
        key = ("massof", id(V_space))
        hit = _adhoc.lookup(key)
        if hit is None or hit[1] is not V_space:
            hit = _adhoc.store(key, (None, V_space, mass_matrix_of(V_space)))     # raises without dolfinx
        V_space = hit[2]
    key = ("mass", id(V_space))
    hit = _adhoc.lookup(key)
    if hit is None or hit[1] is not V_space:
        h = DeviceHierarchy(_options["dim"], 0, 0, c=1, device=_options["device"])
        h.set_flat_level(V_space.tocsr(), prune_zeros=True)
        hit = _adhoc.store(key, (h, V_space, None))
    hit[0].set_vector(0, "v", r)
    return float(np.sqrt(hit[0].quadratic_form(0, "v")))


def err_calculator(u, u_exact, V_space):
    """`sqrt(int (u_h - u_exact)^2)` in the reference (`multigrid.py:213-218`); here `u_exact` is a
    nodal vector and `V_space` a mass matrix or None."""
    return res_calculator(_column(u) - _column(u_exact), V_space)


# ---- the hot path --------------------------------------------------------------------------------------------------------
def _smoother_context(A):
    """Stand-alone smoother on `(D^-1 R, D^-1, level)` operands that do not belong to the initialised
    problem: the two operands go to the device as they are and the sweeps run in the reference's split
    form (`mg_smooth_split`)."""
    key = ("jac", id(A[0]))
    hit = _adhoc.lookup(key)
    if hit is not None and hit[1] is A[0]:
        return hit[0]
    h = DeviceHierarchy(_options["dim"], 0, 0, c=1, device=_options["device"])
    h.set_tuning("require_diagonal", 0)
    h.set_flat_level(A[0].tocsr() if not sp.isspmatrix_csr(A[0]) else A[0])
    h.set_vector(0, "err", A[1].diagonal())
    _adhoc.store(key, (h, A[0], None))
    return h


def jacobiRelaxation(A, v, f, nw):
    """`nw` weighted-Jacobi sweeps (`multigrid.py:223-228`); `v` is not modified."""
    v, f = _column(v), _column(f)
    registered = (A_jacobi_sp_dict is not None and A[2] in A_jacobi_sp_dict
                  and A_jacobi_sp_dict[A[2]][0] is A[0])
    if registered:
        h, level = _hierarchy(), A[2]
    else:
        h, level = _smoother_context(A), 0
        h.set_params(0, 0, omega if omega is not None else 2.0 / 3.0)
    h.set_vector(level, "v", v)
    h.set_vector(level, "f", f)
    if registered:
        h.smooth(level, nw)
    else:
        h.smooth_split(level, nw)
    return h.get_vector(level, "v")


def V_cycle_scheme(A_h, v_h, f_h, test=False):
    """One V(mu1, mu2) cycle from `v_h` (`multigrid.py:231-268`).

    Returns the new `(n, 1)` iterate, or with `test` on the finest level the tuple
    `(v_h, residual_fine_restricted, error_coarse, error_coarse_to_fine_interp)`."""
    h = _hierarchy()
    level = A_h[2]
    h.set_vector(level, "v", _column(v_h))
    h.set_vector(level, "f", _column(f_h))
    h.vcycle(level, 1)
    out = h.get_vector(level, "v")
    if test and level == finest_level and level > coarsest_level:
        return out, h.get_vector(level - 1, "f"), h.get_vector(level - 1, "v"), h.get_vector(level, "err")
    return out


def _fmg(level, f_h, cycles_tol, norm="l2", errors=False):
    h = _hierarchy()
    for l in range(coarsest_level, level):
        h.set_rhs_true(l, b_dict[l])
    h.set_vector(level, "f", _column(f_h))
    if level == coarsest_level:
        h.coarse_solve()
        return h, np.zeros(0)
    return h, h.fmg(mu0, tol=cycles_tol, max_cycles=_options["max_cycles"], top_level=level, norm=norm, errors=errors)


_mass_on_device = None      # (hierarchy, V_fine_dolfx object) whose mass matrix the hierarchy holds


def _finest_norm(h):
    """Which norm `FullMultiGrid` stops on, with the mass matrix handed to the device once: 'mass' = the
    reference's L2(Omega) norm (multigrid.py:203-208, :292-296), 'l2' only when asked for or when no space was given."""
    global _mass_on_device
    if _options["norm"] == "l2" or V_fine_dolfx is None:
        return "l2"
    if _options["norm"] != "auto":
        raise ValueError("configure(norm=...) must be 'auto' or 'l2'")
    if _mass_on_device is None or _mass_on_device[0] is not h or _mass_on_device[1] is not V_fine_dolfx:
        from .dolfinx_adapter import mass_matrix_of
        M = mass_matrix_of(V_fine_dolfx)        # raises TypeError for a space that cannot be assembled here
        h.set_mass(finest_level, M)
        _mass_on_device = (h, V_fine_dolfx)
    return "mass"


def FullMultiGrid(A_h, f_h):
    """Full multigrid (`multigrid.py:271-307`): below the finest level `mu0` V-cycles per level, on the
    finest level V-cycles until the residual norm is <= 1e-11 (`:296`), all on the device (`mg_fmg_ex`: per cycle
    two doubles return to the host).  The norm is the reference's L2(Omega) norm `sqrt(r^T M r)` with the P1 mass
    matrix of the finest mesh: `V_fine_dolfx` is that matrix (SciPy) or a dolfinx function space it is assembled
    from (`dolfinx_adapter.mass_matrix_of`; an error if dolfinx is not importable -- never a silent change of
    norm).  With `V_fine_dolfx = None` or `configure(norm="l2")` it is the l2 norm, which is about 1/h stricter in
    2-D for the same tolerance.  `u_exact_fine` (nodal values) feeds `error_per_V_cycle_finest` every cycle as
    `:292-293` does, in the same norm.  Appends the per-cycle residual norm to `residual_per_V_cycle_finest` and the
    iteration count to `iter_count_for_diff_num_elems_<levels>_levels.csv`, as the reference does (`:295-301`)."""
    level = A_h[2]
    on_finest = level == finest_level
    if on_finest and level > coarsest_level:
        h = _hierarchy()
        norm = _finest_norm(h)
        errors = u_exact_fine is not None and error_per_V_cycle_finest is not None
        if errors:
            h.set_exact(level, _column(u_exact_fine))
        h, out = _fmg(level, f_h, _options["stop_tol"], norm=norm, errors=errors)
        hist, ehist = out if errors else (out, None)
        if errors:
            error_per_V_cycle_finest.extend(float(x) for x in ehist)
        if residual_per_V_cycle_finest is not None:
            residual_per_V_cycle_finest.extend(float(x) for x in hist)
        with open(f'iter_count_for_diff_num_elems_{finest_level - coarsest_level + 1}_levels.csv', mode='a') as fh:
            csv.writer(fh, delimiter=',').writerow(
                [coarsest_level_elements_per_dim * 2 ** finest_level, len(hist)])
    else:
        h, _ = _fmg(level, f_h, 0.0)
    return h.get_vector(level, "v")


def FullMultiGrid_test(A_h, f_h, test=False):
    """FMG with exactly `mu0` cycles on every level (`multigrid.py:312-339`); with `test` returns the
    4-tuple of the last finest cycle, which is what `Multigrid_prototype.py:142-147` prints."""
    level = A_h[2]
    if level == finest_level and level > coarsest_level and not test and _column(f_h).shape[0] != 4:
        # the reference unpacks four values from a single (n, 1) array here (multigrid.py:331-333)
        raise ValueError("too many values to unpack (expected 4)")
    h, _ = _fmg(level, f_h, 0.0)
    out = h.get_vector(level, "v")
    if test and level == finest_level and level > coarsest_level:
        return out, h.get_vector(level - 1, "f"), h.get_vector(level - 1, "v"), h.get_vector(level, "err")
    return out


# ---- CSV helpers (multigrid.py:345-356) ---------------------------------------------------------------------------------------
def writing_residual_for_mesh_to_csv(residual):
    name = (f'residual_for_{coarsest_level_elements_per_dim * 2 ** finest_level}_'
            f'{finest_level - coarsest_level + 1}_levels.csv')
    with open(name, mode='w') as fh:
        w = csv.writer(fh, delimiter=',')
        for i, r in enumerate(residual):
            w.writerow([i, r])


def writing_error_for_mesh_to_csv(error):
    name = (f'error_for_{coarsest_level_elements_per_dim * 2 ** finest_level}_'
            f'{finest_level - coarsest_level + 1}_levels.csv')
    with open(name, mode='w') as fh:
        w = csv.writer(fh, delimiter=',')
        for i, e in enumerate(error):
            w.writerow([i, e])
