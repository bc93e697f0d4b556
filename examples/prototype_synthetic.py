#!/usr/bin/env python3
"""The reference's driver script (`Multigrid_prototype.py`) on the MI355X library.

Same flow as the reference, line for line where dolfinx is not involved:
  parameters                      Multigrid_prototype.py:35-46
  level loop -> A_sp_dict, b_dict Multigrid_prototype.py:62-118   (synthetic dolfinx-convention inputs here,
                                                                    dolfinx itself is not installable offline)
  getJacobiMatrices per level     Multigrid_prototype.py:135-136
  Var_initializer + initialize    Multigrid_prototype.py:138-140
  FullMultiGrid_test(..., True)   Multigrid_prototype.py:141-147  (prints the four shapes)
  FullMultiGrid(...)              Multigrid_prototype.py:148      (commented out in the reference)

    python examples/prototype_synthetic.py            # needs an MI355X and the built libmg_hip.so
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from multigrid_dolfinx_amd import poisson                                            # noqa: E402
from multigrid_dolfinx_amd.multigrid import (FullMultiGrid, FullMultiGrid_test, configure,  # noqa: E402
                                             getJacobiMatrices, initialize_problem)

finest_level = 3
coarsest_level = finest_level - 2          # 3-level V-cycle
coarsest_level_elements_per_dim = 8
mu0, mu1, mu2, omega = 2, 50, 50, 2 / 3

# stands in for the dolfinx level loop: CSR matrices with explicit zeros, lifted right-hand sides and
# coordinate dictionaries, in a non-lexicographic DoF numbering like dolfinx's
parameter = poisson.make_hierarchy(2, coarsest_level, finest_level, c=coarsest_level_elements_per_dim, mu0=mu0,
                                   mu1=mu1, mu2=mu2, omega=omega, seed=0, with_dicts=True)
for key, value in parameter.A_sp_dict.items():
    parameter.A_jacobi_sp_dict[key] = getJacobiMatrices(value)

initialize_problem(parameter)
u_FMG_test, residual_fine_restricted, error_coarse, error_coarse_to_fine_interp = FullMultiGrid_test(
    parameter.A_jacobi_sp_dict[finest_level], parameter.b_dict[finest_level], True)
print(u_FMG_test.shape)
print(residual_fine_restricted.shape)
print(error_coarse.shape)
print(error_coarse_to_fine_interp.shape)

# the production call the reference leaves commented out; the reference's residual reduction per cycle is
# ~0.5-0.65 (injection of the FE residual), so cap the cycle count for the demo
configure(max_cycles=40, stop_tol=1e-11)
initialize_problem(parameter)
u_FMG = FullMultiGrid(parameter.A_jacobi_sp_dict[finest_level], parameter.b_dict[finest_level])
hist = parameter.residual_per_V_cycle_finest
print(f"FullMultiGrid: {len(hist)} V-cycles on the finest level, l2 residual {hist[0]:.3e} -> {hist[-1]:.3e}")
print(f"max |u - u_exact| = {abs(u_FMG - parameter.levels[finest_level].exact()).max():.3e}")
