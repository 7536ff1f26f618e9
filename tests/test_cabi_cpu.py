"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/mldsa_hip.h declares, agrees with the oracle on the parameter table, and fails
loudly (no fallback) when no HIP device is present.  No compute calls here."""
import ctypes as C
import os

import pytest

from conftest import ROOT
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
    spill vector or scalar registers or use scratch memory."""
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
            assert int(v) == 0 and int(sc) == 0, f"{os.path.basename(path)}: {n} spills {v} VGPRs, {sc} B scratch"
            # SGPR "spills" park scalar values in lanes of a VGPR (no memory): harmless in a prologue, v_readlane / v_writelane
            # traffic inside a loop.  k_sign_tail / k_resolve used to spill 32-63: no kernel may spill any
            assert int(sg) == 0, f"{os.path.basename(path)}: {n} spills {sg} SGPRs"
        n_kernels += len(names)
    assert n_kernels >= 60


def test_host_prehash_matches_hashlib(tmp_path):
    """fips204_amd/host/prehash.hpp (the C++ mirror's hash_message, src/hashing.rs:316-354) carries its own SHA-256 /
    SHA-512 / SHAKE128: OID || PH(M) must equal what hashlib gives, across the padding boundaries of all three."""
    import hashlib
    import subprocess
    src = tmp_path / "ph.cpp"
    src.write_text(r'''
#include "fips204_amd/host/prehash.hpp"
#include <cstdio>
using namespace fips204_hip;
int main() {
    const size_t lens[] = {0, 1, 3, 55, 56, 63, 64, 65, 111, 112, 119, 120, 127, 128, 129, 135, 136, 167, 168, 169, 335, 336, 337, 1000, 7727};
    for (size_t n : lens) {
        std::vector<uint8_t> m(n);
        for (size_t i = 0; i < n; i++) m[i] = (uint8_t)(i * 131 + 7 + n);
        for (Ph ph : {Ph::SHA256, Ph::SHA512, Ph::SHAKE128}) {
            const auto v = hash_message(m, ph);
            for (uint8_t b : v) std::printf("%02x", b);
            std::printf("\n");
        }
    }
}
''')
    exe = tmp_path / "ph"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", ROOT, str(src), "-o", str(exe)])
    got = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    from fips204_amd.ml_dsa import hash_message
    oid = bytes([0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02])
    i = 0
    for n in (0, 1, 3, 55, 56, 63, 64, 65, 111, 112, 119, 120, 127, 128, 129, 135, 136, 167, 168, 169, 335, 336, 337, 1000, 7727):
        m = bytes((j * 131 + 7 + n) & 255 for j in range(n))
        want = {"SHA256": oid + b"\x01" + hashlib.sha256(m).digest(), "SHA512": oid + b"\x03" + hashlib.sha512(m).digest(),
                "SHAKE128": oid + b"\x0b" + hashlib.shake_128(m).digest(32)}
        for ph in ("SHA256", "SHA512", "SHAKE128"):
            assert got[i] == want[ph].hex(), (n, ph)
            assert hash_message(m, ph) == want[ph]  # the Python mirror's front-end
            i += 1


def test_offset_tables_are_checked_without_a_device():
    """mldsa_check_offsets (the O(n) pass every *_host entry point makes before it copies by the caller's offsets): pure host
    code of the REAL library, so it runs here.  A decreasing pair -- which would be a ~2^64-byte length -- is MLDSA_ERR_PARAM and
    the message names the entry (the reference's slices carry their own lengths and cannot be malformed, src/traits.rs:330-362)."""
    import numpy as np
    lib = _lib.load()
    def chk(values, n=None):
        a = np.array(values, dtype=np.uint64)
        return lib.mldsa_check_offsets(C.c_void_p(a.ctypes.data), len(values) - 1 if n is None else n)
    assert chk([0]) == 0 and chk([7, 7, 7, 9]) == 0 and chk([0, 5, 5, 2 ** 63, 2 ** 64 - 1]) == 0
    assert lib.mldsa_check_offsets(None, 0) == 0
    assert lib.mldsa_check_offsets(None, 3) == _lib.ERR_PARAM
    assert chk([0, 4, 3, 9]) == _lib.ERR_PARAM and b"entry 2" in lib.mldsa_last_error()
    assert chk([5, 4]) == _lib.ERR_PARAM
    assert chk([0, 2 ** 64 - 8, 16]) == _lib.ERR_PARAM         # a wrapping pair
    assert chk([0, 10, 2 ** 40, 20, 30]) == _lib.ERR_PARAM     # an overshooting entry is caught where the table comes back down
    big = np.arange(1 << 20, dtype=np.uint64)
    big[777_777] = 5
    assert lib.mldsa_check_offsets(C.c_void_p(big.ctypes.data), big.size - 1) == _lib.ERR_PARAM


def test_abi_version_and_sized_stats():
    """mldsa_stats grew by a field; a client built against the shorter struct must not be overrun.
    mldsa_get_stats_sized writes at most the caller's size (checked here with a NULL context: argument errors come first)."""
    lib = _lib.load()
    assert lib.mldsa_abi_version() == _lib.ABI_VERSION
    text = open(_lib.HEADER_PATH).read()
    assert f"#define MLDSA_ABI_VERSION {_lib.ABI_VERSION}" in text
    buf = (C.c_ubyte * 16)()
    assert lib.mldsa_get_stats_sized(None, buf, 16) == _lib.ERR_PARAM
    assert C.sizeof(_lib.Stats) == 48


def test_header_is_plain_c_and_links(tmp_path):
    """The boundary is a C ABI (`extern "C"`, plain pointers and sizes): include/mldsa_hip.h compiles as strict C99 and
    as C++11, and a C program that takes the address of EVERY declared entry point links against libmldsa_hip.so and runs the two
    calls that need no device (mldsa_abi_version, mldsa_check_offsets)."""
    import subprocess
    names = _lib.declared_symbols()
    src = tmp_path / "all_symbols.c"
    body = "\n".join(f"    p[{i}] = (void (*)(void))&{n};" for i, n in enumerate(names))
    src.write_text(f'''#include <stdio.h>
#include "mldsa_hip.h"
int main(void) {{
    void (*p[{len(names)}])(void);
{body}
    for (int i = 0; i < {len(names)}; i++) if (!p[i]) return 2;
    uint64_t good[3] = {{0, 5, 5}}, bad[3] = {{0, 5, 4}};
    if (mldsa_abi_version() != MLDSA_ABI_VERSION) return 3;
    if (mldsa_check_offsets(good, 2) != MLDSA_OK || mldsa_check_offsets(bad, 2) != MLDSA_ERR_PARAM) return 4;
    mldsa_params pr;
    if (mldsa_get_params(MLDSA_65, &pr) != MLDSA_OK || pr.sig_len != 3309 || mldsa_get_params(1, &pr) != MLDSA_ERR_PARAM) return 5;
    printf("%d symbols\\n", {len(names)});
    return 0;
}}
''')
    inc = os.path.join(ROOT, "include")
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "all_symbols"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-Wno-pedantic", "-I", inc, str(src), "-o", str(exe), f"-L{libdir}", "-lmldsa_hip",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == f"{len(names)} symbols", (out.returncode, out.stdout, out.stderr[-500:])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c++", os.path.join(inc, "mldsa_hip.h")])
