"""Builds and runs the C++ host-mirror test (tests/cpp/test_host_mirror.cpp) on the GPU and checks
its messages.rs output against the reference's golden hex (tests/messages.rs:18-20)."""
import os
import subprocess

import pytest

from chacha8rng import ChaCha8Rng

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror(tmp_path, ref_hex):
    exe = tmp_path / "test_host_mirror"
    libdir = os.path.join(ROOT, "fips204_amd", "csrc")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_host_mirror.cpp"), "-o", str(exe),
                           f"-L{libdir}", "-lmldsa_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    rng = ChaCha8Rng(123)
    xi, rnd = rng.fill_bytes(32), rng.fill_bytes(32)
    out = subprocess.run([str(exe), xi.hex(), rnd.hex()], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = dict(ln.split(" ", 1) for ln in out.stdout.splitlines() if " " in ln)
    v = ref_hex["messages_rs"]
    assert lines["sk"] == v["sk"] and lines["sig"] == v["sig"] and lines["pk"] == v["pk"]
    assert out.stdout.strip().endswith("OK")
    # HashML-DSA lines: the mirror's own SHA-256 / SHA-512 / SHAKE128 + MLDSA_MODE_PREHASH against the oracle's
    # restatement of try_hash_sign (src/lib.rs:310-342) with hashlib's hashes
    from oracle import oracle as orc
    sk_o = orc.sk_try_from_bytes(44, bytes.fromhex(lines["sk"]))
    for ph in ("SHA256", "SHA512", "SHAKE128"):
        assert lines["hsig_" + ph] == orc.hash_sign(44, sk_o, b"asdf", rnd, b"ctx", ph).hex(), ph


def test_plain_c_host_example(tmp_path):
    """tests/cpp/c_host.c: a C99 program over the C ABI alone (no C++, no Python) -- keygen, sign, verify through the host-memory entry
    points, one damaged signature, a malformed offset table through both kinds of call, the residue probe."""
    exe = tmp_path / "c_host"
    libdir = os.path.join(ROOT, "fips204_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "c_host.c"), "-o", str(exe), f"-L{libdir}", "-lmldsa_hip", f"-Wl,-rpath,{libdir}",
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), (out.returncode, out.stdout[-800:], out.stderr[-800:])
    assert "verified 99 of 100" in out.stdout and "secret residue: 0 non-zero bytes" in out.stdout
