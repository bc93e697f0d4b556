"""The dolfinx adapter against stand-in objects that expose exactly the methods the reference's driver calls
(`getValuesCSR`, `tabulate_dof_coordinates`, `dofmap.index_map.size_local`, `.array`); dolfinx itself is not
installable here.  CPU part: marshalling only.  GPU part: the adapted bag drives the HIP V-cycle."""
from types import SimpleNamespace

import numpy as np
import pytest

from multigrid_dolfinx_amd import dolfinx_adapter, poisson


class FakeMat:                      # PETSc.Mat stand-in
    def __init__(self, A):
        self._A = A

    def getValuesCSR(self):
        return self._A.indptr.copy(), self._A.indices.copy(), self._A.data.copy()


class FakeSpace:                    # dolfinx.FunctionSpace stand-in (with a ghost row that must be ignored)
    def __init__(self, coords, gdim3=True):
        pad = np.vstack([coords, coords[:1] + 9.0])
        self._c = pad if gdim3 else pad[:, :2]
        self.dofmap = SimpleNamespace(index_map=SimpleNamespace(size_local=coords.shape[0]), index_map_bs=1)

    def tabulate_dof_coordinates(self):
        return self._c


def _levels(seed):
    h = poisson.make_hierarchy(2, 1, 3, seed=seed)
    return h, {l: (FakeMat(L.A), FakeSpace(L.coords, gdim3=(l % 2 == 0)), SimpleNamespace(array=L.b.ravel()))
               for l, L in h.levels.items()}


def test_adapter_reproduces_generator_conventions():
    h, lv = _levels(3)
    bag, gi = dolfinx_adapter.bag_from_dolfinx(lv, 8)
    for l, L in h.levels.items():
        A = bag.A_sp_dict[l][0]
        assert A.indices.dtype == np.int32 and A.indptr.dtype == np.int32 and A.nnz == L.A.nnz
        assert np.array_equal(A.data, L.A.data) and np.array_equal(A.indices, L.A.indices)
        assert np.array_equal(gi[l], L.grid_index)
        assert bag.b_dict[l].shape == (L.n, 1) and np.array_equal(bag.b_dict[l], L.b)
    with pytest.raises(TypeError):
        dolfinx_adapter.csr_from_petsc(FakeMat(h.levels[1].A.astype(np.complex128)))
    with pytest.raises(ValueError):
        dolfinx_adapter.level_from_dolfinx(FakeMat(h.levels[1].A), h.levels[2].coords)


@pytest.mark.gpu
def test_adapted_hierarchy_drives_the_device_v_cycle():
    from multigrid_dolfinx_amd import multigrid as mg
    from oracle.mg_oracle import Oracle
    h, lv = _levels(5)
    bag, gi = dolfinx_adapter.bag_from_dolfinx(lv, 8, mu1=3, mu2=3)
    mg.configure(dim=2, grid_index=gi)
    try:
        mg.initialize_problem(bag)
        f = bag.b_dict[3]
        got = mg.V_cycle_scheme(bag.A_jacobi_sp_dict[3], np.zeros_like(f), f)
    finally:
        mg.configure(grid_index=None)
    h.mu1 = h.mu2 = 3
    orc = Oracle(h, {l: L.grid_index for l, L in h.levels.items()}, dim=2)
    want = orc.v_cycle(orc.A_jacobi_sp_dict[3], np.zeros_like(f), f)
    assert np.linalg.norm(got - want) <= 1e-10 * np.linalg.norm(want)
