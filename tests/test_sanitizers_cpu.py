"""Sanitizer leg (the reference fuzzes with debug-assertions + overflow-checks, fuzz/Cargo.toml:31-35).
GPU AddressSanitizer does not exist on the pool, so the CPU-side code runs under ASan + UBSan here:

  * the oracle (oracle/mldsa_oracle.c built as liboracle_asan.so, oracle/Makefile) through the reference's ACVP KATs and
    through arbitrary / out-of-range key bytes, in a subprocess with the ASan runtime preloaded;
  * the C++ host mirror (fips204_amd/host/fips204_hip.hpp: argument packing, RAII device buffers, group calls) and its
    SHA-256 / SHA-512 / SHAKE128 (prehash.hpp) over a stub C ABI whose buffers are exactly sized (tests/cpp/stub_cabi.cpp);
  * the REAL csrc/batcher.cpp (host code: queues, spinlock, futex wake-ups, key table) over stand-ins for the device
    (stub_cabi.cpp, stub_hip.cpp) under ThreadSanitizer, and once more under ASan + UBSan.
"""
import hashlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    out = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def test_cpp_mirror_under_asan_ubsan(tmp_path):
    exe = tmp_path / "test_mirror_asan"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wall",
                           os.path.join(ROOT, "tests", "cpp", "test_mirror_asan.cpp"), os.path.join(ROOT, "tests", "cpp", "stub_cabi.cpp"),
                           "-o", str(exe)])
    oid = bytes([0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02])
    lines = []
    for n in (0, 1, 3, 55, 56, 63, 64, 65, 111, 112, 119, 120, 127, 128, 129, 167, 168, 169, 335, 336, 337, 1000, 5000):
        m = bytes((i * 7 + n) & 255 for i in range(n))
        for ph, suffix, h in (("SHA256", b"\x01", lambda x: hashlib.sha256(x).digest()), ("SHA512", b"\x03", lambda x: hashlib.sha512(x).digest()),
                              ("SHAKE128", b"\x0b", lambda x: hashlib.shake_128(x).digest(32))):
            lines.append(f"{ph} {m.hex() or '-'} {(oid + suffix + h(m)).hex()}")
    out = subprocess.run([str(exe)], input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    assert out.stdout.startswith("OK 69 hash vectors"), out.stdout


@pytest.mark.parametrize("sanitizer,extra", [("thread", ()), ("address,undefined", ()), ("address,undefined", ("-DMLDSA_BATCHER_PORTABLE",)),
                                             ("address,undefined", ("-DMLDSA_TEST_NO_ZEROISE",))])
def test_batcher_under_sanitizers(tmp_path, sanitizer, extra):
    """csrc/batcher.cpp is plain C++ over the C ABI and a dozen HIP runtime calls: built here with g++ against stand-ins for both,
    24 threads of random single-operation calls (one and three lanes, 40 keys over 16 table slots, batches of 8), every result
    compared with the batched stand-in's.  ThreadSanitizer sees every access to the shared state (the synchronisation is C++
    atomics; the futex calls only park threads); ASan + UBSan see the staging arithmetic.  Beside the callers a thread forgets keys,
    flushes the tables and switches the private-key cache off and on.  Every buffer the batcher releases (page-locked staging, device
    memory) is scanned for private-key bytes by a hook of the stand-ins: none in the library build, some in the build whose wipe_host does
    nothing (-DMLDSA_TEST_NO_ZEROISE: the probe's negative control).  -DMLDSA_BATCHER_PORTABLE: the condition-variable fallback of the two
    futex calls (what a non-Linux host gets) passes the same run."""
    if sanitizer == "thread" and not _runtime("libtsan.so"):
        pytest.skip("no libtsan.so next to gcc")
    exe = tmp_path / "batcher_san"
    src = [os.path.join(ROOT, "fips204_amd", "csrc", "batcher.cpp")] + [os.path.join(ROOT, "tests", "cpp", f) for f in
                                                                        ("stub_cabi.cpp", "stub_hip.cpp", "test_batcher_tsan.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-fno-sanitize-recover=all", "-pthread", "-D__HIP_PLATFORM_AMD__",
                           "-I/opt/rocm/include", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "fips204_amd", "csrc")] + list(extra) + src + ["-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1", ASAN_OPTIONS="detect_leaks=1",
                                  UBSAN_OPTIONS="print_stacktrace=1"))
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), (out.stdout[-1000:], out.stderr[-4000:])
    assert "ThreadSanitizer" not in out.stderr and "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]


_ARBITRARY_KEYS = r"""
import numpy as np
from oracle import oracle as orc
rng = np.random.default_rng(5)
orc.lib()
maps = open("/proc/self/maps").read()
assert "liboracle_asan.so" in maps and "libasan" in maps, "the sanitizer build is not what is loaded"
for pset in (44, 65, 87):
    p = orc.params(pset)
    for trial in range(6):
        skb = rng.integers(0, 256, p.sk_len, dtype=np.uint8).tobytes()
        if trial == 0: skb = skb[:128] + b"\xff" * (p.sk_len - 128)      # every field all-ones: eta / t0 out of range
        if trial == 1: skb = skb[:128] + b"\x00" * (p.sk_len - 128)
        sk = orc.sk_try_from_bytes(pset, skb)
        pk = orc.get_public_key(pset, sk)
        assert len(orc.pk_into_bytes(pset, pk)) == p.pk_len and len(orc.sk_into_bytes(pset, sk)) == p.sk_len
        msg = bytes(range(trial * 9))
        sig = orc.sign_internal(pset, sk, msg, bytes(32), ctx=b"c" * trial, mode=0)
        orc.verify_internal(pset, pk, msg, sig, ctx=b"c" * trial, mode=0)
        # arbitrary public-key and signature bytes (fuzz_all.rs:25-37): every failure is just False
        pk2 = orc.pk_try_from_bytes(pset, rng.integers(0, 256, p.pk_len, dtype=np.uint8).tobytes())
        assert orc.verify_internal(pset, pk2, msg, sig, mode=0) is False
        assert orc.verify_internal(pset, pk, msg, rng.integers(0, 256, p.sig_len, dtype=np.uint8).tobytes(), mode=0) is False
    try:
        orc.sign_internal(pset, sk, b"m", bytes(32), ctx=b"x" * 256, mode=0)
        raise SystemExit("ctx of 256 bytes accepted")
    except ValueError:
        pass
print("arbitrary keys OK")
"""


def test_oracle_kats_and_arbitrary_keys_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if not asan:
        pytest.skip("no libasan.so next to gcc")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               MLDSA_ORACLE_LIB=os.path.join(ROOT, "oracle", "liboracle_asan.so"), PYTHONPATH=ROOT)
    # the reference's ACVP keyGen / sigGen / sigVer vectors, messages.rs, bad_sig and the unit pins through the sanitizer build
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_kat.py"), "-x", "-q", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "passed" in out.stdout and "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
    out = subprocess.run([sys.executable, "-c", _ARBITRARY_KEYS], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0 and "arbitrary keys OK" in out.stdout, (out.stdout[-2000:], out.stderr[-3000:])
    assert "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]


# ------------------------------------------------------------------------------ the library's host orchestration code
CSRC = os.path.join(ROOT, "fips204_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
HOST_SOURCES = ["capi.hip", "pipeline.hip", "host_api.hip", "group.hip", "kernels_poly.hip", "kernels_sample.hip", "kernels_codec.hip", "kernels_sign.hip",
                "kernels_small.hip", "batcher.cpp", "tables.cpp"]


def _build_host_fuzz(out_dir, sanitizer, replace=None, reuse=None):
    """hipcc --offload-host-only -fsanitize=<sanitizer> over every source of the library + tests/cpp/stub_hip_full.cpp + tests/cpp/fuzz_host.cpp.
    replace = {source name: path of a mutated copy}; reuse = a directory built before whose other objects are taken as they are."""
    os.makedirs(out_dir, exist_ok=True)
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "--offload-host-only", f"-fsanitize={sanitizer}", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
             "-Wno-unused-function", "-Wno-option-ignored", "-I", CSRC, "-I", os.path.join(ROOT, "include"), "-x", "hip"]
    jobs = []
    sources = [(n, os.path.join(CSRC, n)) for n in HOST_SOURCES] + [(n, os.path.join(ROOT, "tests", "cpp", n)) for n in ("stub_hip_full.cpp", "fuzz_host.cpp")]
    for name, path in sources:
        obj = os.path.join(out_dir, name.rsplit(".", 1)[0] + ".o")
        if replace and name in replace:
            path = replace[name]
        elif reuse:
            import shutil
            shutil.copy2(os.path.join(reuse, os.path.basename(obj)), obj)
            continue
        jobs.append((name, subprocess.Popen([HIPCC] + flags + ["-c", path, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for name, j in jobs:
        out, _ = j.communicate(timeout=900)
        assert j.returncode == 0, (name, out[-3000:])
    objs = sorted(os.path.join(out_dir, f) for f in os.listdir(out_dir) if f.endswith(".o") and f != "fatbins.o")
    # host-only objects still name the device code objects they would have carried: empty stand-ins
    nm = subprocess.run(["nm", "-u"] + objs, capture_output=True, text=True).stdout
    import re
    fat = sorted(set(re.findall(r"__hip_fatbin_[0-9a-f]+", nm)))
    with open(os.path.join(out_dir, "fatbins.c"), "w") as f:
        f.write("".join(f"char {s}[8];\n" for s in fat))
    subprocess.check_call(["gcc", "-c", os.path.join(out_dir, "fatbins.c"), "-o", os.path.join(out_dir, "fatbins.o")])
    exe = os.path.join(out_dir, "fuzz_host")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", f"-fsanitize={sanitizer}", "-o", exe] + objs + [os.path.join(out_dir, "fatbins.o"), "-lpthread", "-ldl"])
    return exe


def _run_fuzz(exe, seconds, seed, threads, timeout=900):
    return subprocess.run([exe, str(seconds), str(seed), str(threads)], capture_output=True, text=True, timeout=timeout,
                          env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
                                   TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"))


def _clean(out):
    return (out.returncode == 0 and out.stdout.strip().endswith("OK") and "AddressSanitizer" not in out.stderr and "ThreadSanitizer" not in out.stderr
            and "runtime error" not in out.stderr)


@pytest.fixture(scope="module")
def host_fuzz_asan(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    return _build_host_fuzz(str(tmp_path_factory.mktemp("host_asan")), "address,undefined")


def test_host_orchestration_under_asan_ubsan(host_fuzz_asan):
    """capi / pipeline / host_api / group .hip + the kernel launchers + batcher.cpp, compiled host-only with ASan + UBSan and linked over a
    stand-in HIP runtime whose device memory is exactly-sized heap memory (tests/cpp/stub_hip_full.cpp), driven by tests/cpp/fuzz_host.cpp:
    random (n_ops, set, mode, options, workspace caps, caller-owned workspaces, malformed offset tables, out-of-range key indices, groups
    of 1 ... 8 contexts, graph-cache eviction, a device that runs out of memory) from two threads on a shared context, a fixed seed list,
    MLDSA_FUZZ_SECONDS in all (default 60).  The reference's counterpart: fuzz/fuzz_targets/fuzz_all.rs:20-51 under debug-assertions +
    overflow-checks (fuzz/Cargo.toml:31-35).  First run of this leg found a use-after-free: mldsa_pk_expand / mldsa_sk_expand touched the
    context's profiling marks without its mutex."""
    total = float(os.environ.get("MLDSA_FUZZ_SECONDS", "60"))
    seeds = (20261003, 7, 99)
    calls = 0
    for seed in seeds:
        out = _run_fuzz(host_fuzz_asan, total / len(seeds), seed, 2)
        assert _clean(out), (seed, out.stdout[-1500:], out.stderr[-6000:])
        import re
        m = re.search(r"(\d+) calls .* (\d+) kernel launches, (\d+) touch-modelled, (\d+) allocations refused", out.stdout)
        assert m and int(m.group(2)) > 500 and int(m.group(3)) > 100, out.stdout
        calls += int(m.group(1))
    assert calls > 300, calls


def test_host_orchestration_under_tsan(tmp_path):
    """the same driver, four threads (a shared context, private ones, groups with their worker threads, batcher-free), under ThreadSanitizer"""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    exe = _build_host_fuzz(str(tmp_path / "host_tsan"), "thread")
    out = _run_fuzz(exe, float(os.environ.get("MLDSA_FUZZ_TSAN_SECONDS", "25")), 31337, 4)
    assert _clean(out), (out.stdout[-1500:], out.stderr[-6000:])


@pytest.mark.parametrize("what,old,new,expect", [
    # reserve_workspace asks for one byte less than the pass needs: the pipelines' own "workspace not reserved" checks refuse calls that
    # nothing bounds -- the driver reports an unexpected MLDSA_ERR_NOMEM
    ("reserve_workspace", "const int rc = ensure_workspace(ctx, bytes);", "const int rc = ensure_workspace(ctx, bytes - 1);", "workspace not reserved"),
    # ensure_workspace allocates one byte less than it records: the clearing of a signing pass's secrets runs off the allocation -- ASan
    ("ensure_workspace", "hipError_t e = malloc_quiesced(&ctx->ws, bytes);", "hipError_t e = malloc_quiesced(&ctx->ws, bytes - 1);", "heap-buffer-overflow"),
])
def test_a_planted_off_by_one_in_the_workspace_code_turns_the_leg_red(host_fuzz_asan, tmp_path, what, old, new, expect):
    """negative control of the leg above: the same build with ONE mutated line of pipeline.hip"""
    src = open(os.path.join(CSRC, "pipeline.hip")).read()
    assert src.count(old) == 1, what
    mutated = tmp_path / "pipeline_mutated.hip"
    mutated.write_text(src.replace(old, new))
    exe = _build_host_fuzz(str(tmp_path / "planted"), "address,undefined", replace={"pipeline.hip": str(mutated)}, reuse=os.path.dirname(host_fuzz_asan))
    out = _run_fuzz(exe, 30, 20261003, 2)
    assert not _clean(out), out.stdout[-500:]
    assert expect in out.stderr, (what, out.stdout[-800:], out.stderr[-3000:])


def test_the_stand_ins_copies_of_kernel_argument_structs_match_the_sources():
    """tests/cpp/stub_hip_full.cpp reads two by-value kernel arguments whose types live in .hip files: same fields, same order"""
    import re
    stub = open(os.path.join(ROOT, "tests", "cpp", "stub_hip_full.cpp")).read()

    def fields(text, name):
        body = re.search(r"struct %s \{(.*?)\};" % name, text, flags=re.S).group(1)
        body = re.sub(r"//[^\n]*", "", body)
        return [" ".join(x.replace("*", " * ").split()) for x in body.split(";") if x.strip()]
    assert fields(stub, "VerdictArgs") == fields(open(os.path.join(CSRC, "kernels_codec.hip")).read(), "VerdictArgs")
    assert fields(stub, "KeygenOut") == fields(open(os.path.join(CSRC, "kernels_poly.hip")).read(), "KeygenOut")
