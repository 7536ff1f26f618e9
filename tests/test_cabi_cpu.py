"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/mldsa_hip.h declares, agrees with the oracle on the parameter table, and fails
loudly (no fallback) when no HIP device is present.  No compute calls here."""
import ctypes as C

import pytest

from fips204_amd import _lib
from oracle import oracle as orc


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    declared = _lib.declared_symbols()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/mldsa_hip.h but not exported"
    assert set(_lib._SIGNATURES) == set(declared)


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_param_table_matches_reference(pset):  # src/lib.rs:639-740
    p, o = _lib.get_params(pset), orc.params(pset)
    for f, _ in _lib.Params._fields_:
        assert getattr(p, f) == getattr(o, f), f


def test_unknown_param_set_is_an_error():
    with pytest.raises(_lib.MldsaError):
        _lib.get_params(99)


def test_no_device_fails_loudly():
    lib = _lib.load()
    if lib.mldsa_device_count() > 0:
        pytest.skip("a HIP device is present")
    h = C.c_void_p()
    rc = lib.mldsa_ctx_create(0, C.byref(h))
    assert rc < 0 and not h.value
    assert lib.mldsa_last_error()


def test_product_never_imports_oracle():
    import os
    import re
    root = os.path.dirname(os.path.abspath(_lib.__file__))
    for dirpath, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f
                assert "liboracle" not in text and "mldsa_oracle" not in text, f
