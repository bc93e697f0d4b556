"""ctypes binding of `libmg_hip.so` (C ABI: `include/mg_hip.h`).

There is deliberately no fallback: if the shared library is missing, or no MI355X is
visible, every entry point raises.  Build with `python __graft_entry__.py` (or
`make -C multigrid_dolfinx_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmg_hip.so")

MG_VEC_V, MG_VEC_F, MG_VEC_R, MG_VEC_ERR = 0, 1, 2, 3
MG_RESTRICT_INJECTION, MG_RESTRICT_FULL_WEIGHTING, MG_RESTRICT_TABLE = 0, 1, 2
MG_SMOOTH_JACOBI, MG_SMOOTH_RBGS, MG_SMOOTH_MCGS = 0, 1, 2
MG_NORM_L2, MG_NORM_MASS = 0, 1

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)
ALLGATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64,
                            C.POINTER(C.c_double), C.POINTER(C.c_int64))

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_i64p = C.POINTER(C.c_int64)
_H = C.c_void_p

# name -> (argtypes); every function returns int status except mg_last_error
SIGNATURES = {
    "mg_create": [C.c_int, C.c_int, C.c_int, C.POINTER(_H)],
    "mg_destroy": [_H],
    "mg_device_info": [_H, C.c_char_p, C.c_size_t],
    "mg_comm_unique_id": [C.c_void_p, C.c_size_t],
    "mg_set_comm": [_H, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int64],
    "mg_comm_selftest": [C.c_int],
    "mg_set_comm_callbacks": [_H, C.c_int, C.c_int, EXCHANGE_FN, ALLREDUCE_FN, ALLGATHERV_FN, C.c_void_p, C.c_int64],
    "mg_set_level_csr": [_H, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                         C.c_void_p, C.c_int],
    "mg_set_level_grid": [_H, C.c_int, C.c_int, C.c_int64, C.c_void_p],
    "mg_level_slab": [_H, C.c_int, C.c_int, _i64p, _i64p, _i64p, _i64p],
    "mg_set_level_csr_local": [_H, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_int],
    "mg_gen_poisson_level": [_H, C.c_int, C.c_int, C.c_int],
    "mg_gen_lattice_level": [_H, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    "mg_jacobi_split": [C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                        C.c_void_p, C.c_void_p],
    "mg_set_params": [_H, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int],
    "mg_set_tuning": [_H, C.c_char_p, C.c_int64],
    "mg_set_prolongation_table": [_H, C.c_void_p, C.c_void_p, C.c_void_p],
    "mg_set_restriction_table": [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p],
    "mg_level_info": [_H, C.c_int, _i64p, _i64p, _i64p, _i64p, _i64p, _ip, _ip, _ip],
    "mg_level_row_classes": [_H, C.c_int, _ip],
    "mg_level_storage": [_H, C.c_int, _ip, _i64p, _i64p, _ip, _ip, _i64p],
    "mg_set_vector": [_H, C.c_int, C.c_int, C.c_void_p],
    "mg_get_vector": [_H, C.c_int, C.c_int, C.c_void_p, C.c_int],
    "mg_zero_vector": [_H, C.c_int, C.c_int],
    "mg_copy_vector": [_H, C.c_int, C.c_int, C.c_int],
    "mg_smooth": [_H, C.c_int, C.c_int],
    "mg_smooth_split": [_H, C.c_int, C.c_int],
    "mg_residual": [_H, C.c_int],
    "mg_restrict": [_H, C.c_int, C.c_int],
    "mg_prolong": [_H, C.c_int, C.c_int],
    "mg_coarse_solve": [_H, _ip, _dp],
    "mg_vcycle": [_H, C.c_int, C.c_int, C.c_void_p],
    "mg_prepare_cycle": [_H, C.c_int],
    "mg_norm2": [_H, C.c_int, C.c_int, _dp],
    "mg_quadratic_form": [_H, C.c_int, C.c_int, _dp],
    "mg_set_rhs_true": [_H, C.c_int, C.c_void_p],
    "mg_fmg": [_H, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p, _ip],
    "mg_set_mass_csr": [_H, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p],
    "mg_set_exact": [_H, C.c_int, C.c_void_p],
    "mg_fmg_ex": [_H, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p, _ip],
    "mg_counters": [_H, _i64p, _i64p, _i64p, _ip],
    "mg_time_kernel": [_H, C.c_char_p, C.c_int, C.c_int, _dp],
    "mg_sync": [_H],
    "mg_memory_bytes": [_H, _i64p],
}

_lib = None


class MgError(RuntimeError):
    """A non-zero status from libmg_hip.so; the message is mg_last_error()."""


def build_extension():
    """Compile libmg_hip.so in-tree with hipcc (gfx950).  Building is not a fallback: nothing runs without it."""
    import subprocess
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc")], check=True)


def load():
    """Load libmg_hip.so and declare every prototype of include/mg_hip.h."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MgError(
            f"{LIB_PATH} is missing: build the HIP extension first (python __graft_entry__.py). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.mg_last_error.argtypes = []
    lib.mg_last_error.restype = C.c_char_p
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise MgError(load().mg_last_error().decode("utf-8", "replace"))


def as_f64(x, n=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))
    if n is not None and a.size != n:
        raise ValueError(f"vector has {a.size} entries, level has {n}")
    return a


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)
