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


def test_no_kernel_spills_registers():
    """k_verify_main<8,7,...,APACK> spilled 8 VGPRs.  The Makefile keeps hipcc's
    -Rpass-analysis=kernel-resource-usage remarks next to every object (*.res): no instantiated kernel may
    spill vector registers or use scratch memory."""
    import glob
    import os
    import re
    from fips204_amd import build
    csrc = os.path.join(os.path.dirname(os.path.abspath(_lib.__file__)), "csrc")
    if not glob.glob(os.path.join(csrc, "kernels_*.res")):
        build.build(force=True)
    n_kernels = 0
    for path in sorted(glob.glob(os.path.join(csrc, "*.res"))):
        text = open(path).read()
        names = re.findall(r"Function Name: (\S+)", text)
        spills = re.findall(r"VGPRs Spill: (\d+)", text)
        sgpr_spills = re.findall(r"SGPRs Spill: (\d+)", text)
        scratch = re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", text)
        assert len(names) == len(spills) == len(scratch) == len(sgpr_spills)
        for n, v, sg, sc in zip(names, spills, sgpr_spills, scratch):
            # (SGPR "spills" are allowed: the compiler parks scalar values in lanes of a VGPR, no memory is involved)
            assert int(v) == 0 and int(sc) == 0, f"{os.path.basename(path)}: {n} spills {v} VGPRs, {sc} B scratch"
        n_kernels += len(names)
    assert n_kernels >= 60
