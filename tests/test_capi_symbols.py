"""CPU checks of the C-ABI boundary: the library loads without a GPU, exports every symbol
`include/mg_hip.h` declares, and refuses to compute without a device (no CPU fallback)."""
import os
import re

import pytest

from multigrid_dolfinx_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mg_[a-z0-9_]+)\s*\(", text)) - {"mg_exchange_fn", "mg_allreduce_fn",
                                                                        "mg_allgatherv_fn"})


def test_header_and_binding_agree():
    names = declared_functions()
    assert "mg_vcycle" in names and "mg_last_error" in names
    assert set(names) == set(_capi.SIGNATURES) | {"mg_last_error"}


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    for name in declared_functions():
        assert getattr(lib, name) is not None, name


def test_no_cpu_fallback_without_a_device():
    import subprocess, sys
    code = ("import ctypes as C, sys; sys.path.insert(0, %r);"
            "from multigrid_dolfinx_amd import _capi; lib=_capi.load(); h=C.c_void_p();"
            "rc=lib.mg_create(3,2,0,C.byref(h)); print(rc, lib.mg_last_error().decode())" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr
    rc, msg = out.stdout.strip().split(" ", 1)
    assert rc != "0" and msg          # fails loudly, with a message


def test_missing_library_raises(monkeypatch):
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", os.path.join(ROOT, "does_not_exist.so"))
    with pytest.raises(_capi.MgError):
        _capi.load()


def test_every_tuning_key_is_documented_in_the_header():
    """`mg_set_tuning` takes its keys as strings: the header is their only documentation.  Every key the library
    accepts must be listed there and every key listed there must be accepted."""
    src = open(os.path.join(ROOT, "multigrid_dolfinx_amd", "csrc", "mg_capi.hip")).read()
    body = src[src.index("int mg_set_tuning("):]
    body = body[:body.index("\nnamespace {")]
    accepted = set(re.findall(r'k == "([a-z_0-9]+)"', body))
    header = open(os.path.join(ROOT, "include", "mg_hip.h")).read()
    block = header[header.index("Tuning and format knobs"):header.index("int mg_set_tuning")]
    documented = set(re.findall(r'^\s*\*\s+"([a-z_0-9]+)"', block, flags=re.M))
    assert accepted and accepted == documented, (sorted(accepted - documented), sorted(documented - accepted))
