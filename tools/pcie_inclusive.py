"""Host-buffer (PCIe-inclusive) vs device-resident V-cycle rate through the drop-in shim; the number quoted in DESIGN.md section 2."""
import sys, time, numpy as np
sys.path.insert(0, '.')
from multigrid_dolfinx_amd import poisson
from multigrid_dolfinx_amd import multigrid as mg
dim, lo, hi = 3, 2, 4
bag = poisson.make_hierarchy(dim, lo, hi, c=8, mu1=50, mu2=50)
for l in bag.A_sp_dict: bag.A_jacobi_sp_dict[l] = (None, None, l)
mg.configure(dim=3, grid_index={l: L.grid_index for l, L in bag.levels.items()})
mg.initialize_problem(bag)
f = bag.b_dict[hi]; v = np.zeros_like(f)
v = mg.V_cycle_scheme(bag.A_jacobi_sp_dict[hi], v, f)   # builds hierarchy
t=time.perf_counter()
for _ in range(5): v = mg.V_cycle_scheme(bag.A_jacobi_sp_dict[hi], v, f)
t_host=(time.perf_counter()-t)/5
h = mg._hierarchy()
h.set_vector(hi,'v',np.zeros_like(f)); h.set_vector(hi,'f',f); h.vcycle(hi,1); h.sync()
t=time.perf_counter(); h.vcycle(hi,5); h.sync(); t_dev=(time.perf_counter()-t)/5
print(f"3-D N=128 ({f.size} DoF) V(50,50): host-buffer V_cycle_scheme {t_host*1e3:.2f} ms/cycle ({1/t_host:.1f}/s), device-resident {t_dev*1e3:.2f} ms/cycle ({1/t_dev:.1f}/s)")
