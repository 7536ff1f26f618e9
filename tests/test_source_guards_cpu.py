"""Textual guards over the library's sources: properties a reviewer would otherwise have to re-check by eye after every change."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "fips204_amd", "csrc")


def _function_body(text, signature_start):
    i = text.index(signature_start)
    j = text.index("{", i)
    depth, k = 0, j
    while True:
        c = text[k]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return text[j:k + 1]
        k += 1


def test_the_plan_cache_is_keyed_by_everything_the_plan_reads():
    """plan_sign keeps the last plans per thread (pipeline.hip): every field of the context that plan_sign_compute reads must be part of
    the cache key, or a changed option would be answered with a stale plan."""
    src = open(os.path.join(CSRC, "pipeline.hip")).read()
    compute = _function_body(src, "static SignPlan plan_sign_compute(")
    cached = _function_body(src, "SignPlan plan_sign(const mldsa_ctx *ctx")
    read = set(re.findall(r"ctx->(\w+)", compute))
    keyed = set(re.findall(r"ctx->(\w+)", cached))
    assert read and read <= keyed, sorted(read - keyed)
    # the arguments are in the key as well
    for arg in ("set", "n", "async_mode", "plan_stop"):
        assert re.search(r"k\.[vd]\[\d+\] = [^;]*\b%s\b" % arg, cached), arg


def test_every_lane_of_a_signing_call_has_its_own_arrival_counters():
    """the single-launch kernels hand over through arrival counters; kernel chains that can be in flight at the same time -- the lanes of a
    large signing call -- must not share them (found by the soak in round 5): nothing in the signing pipeline takes the context's array
    directly, every user goes through the lane's own set."""
    src = open(os.path.join(CSRC, "pipeline.hip")).read()
    sign_part = src[src.index("struct SignWs"):]
    uses = [u for u in re.findall(r"[^\n]*ctx->d_small_ctr[^\n]*", sign_part) if not u.split("ctx->d_small_ctr")[0].rstrip().endswith("(") and "//" not in u.split("ctx->d_small_ctr")[0]]
    assert len(uses) == 1 and "small_ctr = ctx->d_small_ctr + (size_t)i * SMALL_CTR_ENTRIES" in uses[0], uses
    ctx_h = open(os.path.join(CSRC, "ctx.h")).read()
    sets = int(re.search(r"constexpr size_t SMALL_CTR_SETS = (\d+);", ctx_h).group(1))
    assert sets >= 2
    assert "SMALL_CTR_SETS * SMALL_CTR_ENTRIES" in open(os.path.join(CSRC, "capi.hip")).read()


def test_the_product_has_no_cpu_fallback_and_no_compat_layer():
    """no hipify shape, no dual CUDA / HIP paths, no Triton; the Python side fails loudly when the library is missing"""
    for root, _, files in os.walk(os.path.join(ROOT, "fips204_amd")):
        for f in files:
            if f.endswith((".hip", ".h", ".cpp", ".py")):
                t = open(os.path.join(root, f), errors="replace").read()
                assert "__HIP_PLATFORM_AMD__" not in t and "__CUDACC__" not in t and "import triton" not in t, f
    lib = open(os.path.join(ROOT, "fips204_amd", "_lib.py")).read()
    assert "raise" in lib and "libmldsa_hip.so" in lib


def test_every_environment_variable_the_library_reads_is_in_the_header_and_gated():
    """VERDICT r5 'What's weak' 10: mldsa_ctx_create read 27 undocumented MLDSA_* variables.  Now: every name the library passes to getenv
    (directly or through env_long) is listed in include/mldsa_hip.h's "Environment" section; the knobs go through env_long, which returns
    the default unless MLDSA_TUNING_ENV=1 (or -DMLDSA_TUNING); the three closed knobs are gone from the code."""
    header = open(os.path.join(ROOT, "include", "mldsa_hip.h")).read()
    env_doc = header[header.index("/* Environment."):]
    env_doc = env_doc[:env_doc.index("*/")]
    names = set()
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h", ".cpp")):
            t = open(os.path.join(CSRC, f)).read()
            names |= set(re.findall(r'getenv\("(\w+)"\)', t)) | set(re.findall(r'env_long\("(\w+)"', t))
            # no other way into the environment
            assert "secure_getenv" not in t and "environ" not in re.sub(r"//[^\n]*", "", t), f
    assert len(names) >= 20 and "MLDSA_TUNING_ENV" in names
    missing = sorted(n for n in names if not re.search(r"\b%s\b" % n, env_doc))
    assert not missing, missing
    # ... and nothing is documented that is not read
    documented = set(re.findall(r"\bMLDSA_[A-Z0-9_]+\b", env_doc)) - {n for n in re.findall(r"\bMLDSA_OPT_[A-Z0-9_]+\b", env_doc)}
    closed = {"MLDSA_SIB_THIRD_STREAM", "MLDSA_SIDE_PROLOGUE", "MLDSA_SPEC_ALPHA"}
    assert documented - closed - {"MLDSA_TUNING"} == names, sorted((documented - closed - {"MLDSA_TUNING"}) ^ names)
    capi = open(os.path.join(CSRC, "capi.hip")).read()
    body = _function_body(capi, "static long env_long(")
    assert body.index("if (!tuning_env_on()) return dflt;") < body.index("getenv(name)")
    # direct getenv calls: the switch itself and the print-only debug variable
    direct = {}
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h", ".cpp")):
            for n in re.findall(r'getenv\("(\w+)"\)', open(os.path.join(CSRC, f)).read()):
                direct[n] = f
    assert set(direct) == {"MLDSA_TUNING_ENV", "MLDSA_DEBUG_IGNORED"}, direct
    for f in os.listdir(CSRC):
        if f.endswith((".hip", ".h", ".cpp")):
            t = open(os.path.join(CSRC, f)).read()
            for gone in ("opt_sib_third", "opt_side_prologue", "opt_spec_alpha"):
                assert gone not in t, (f, gone)
            for gone in closed:
                assert gone not in t, (f, gone)


def test_the_shipped_library_is_built_without_experiment_variants_and_with_late_arguments():
    """field.h MLDSA_EXP selects measured memory-path variants for A/Bs (`make variants`), -DMLDSA_NO_LATE_ARG the fallback for a toolchain
    whose kernarg layout fails the self-test: neither belongs in the default build's flags."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^CXXFLAGS \?= (.*)$", mk, flags=re.M).group(1)
    assert "MLDSA_EXP" not in flags and "MLDSA_NO_LATE_ARG" not in flags and "MLDSA_TUNING" not in flags
    fh = open(os.path.join(CSRC, "field.h")).read()
    assert re.search(r"#ifndef MLDSA_EXP\s*\n#define MLDSA_EXP 0\s*\n#endif", fh)
    # the adopted cache policy is on when MLDSA_EXP is 0, the rejected variants are off
    assert "NT_A_VERIFY = (MLDSA_EXP & 2) == 0" in fh and "NT_ZC = (MLDSA_EXP & 256) == 0" in fh and "NT_A_KG = (MLDSA_EXP & 512) == 0" in fh
    for rejected in ("EXP_NT_A_SIGN", "EXP_NT_STORE", "EXP_LDSDMA", "EXP_NT_DMA"):
        assert re.search(r"constexpr bool %s = \(MLDSA_EXP & \d+\) != 0;" % rejected, fh), rejected


def test_the_lane_policy_the_header_documents_is_the_one_the_pipeline_applies():
    """MLDSA_OPT_SIGN_LANES = 0 (default since round 6): two slices from 131 072 ops on, ML-DSA-44 from 65 536 (profiles/r06_ab_sign_lanes_*).
    The sizes live in pipeline.hip; the header states them; an exporting call stays one lane and the workspace holds either layout."""
    pl = open(os.path.join(CSRC, "pipeline.hip")).read()
    m = re.search(r"sign_lanes_auto_min_ops\(int set\) \{ return set == MLDSA_44 \? (\d+) : (\d+); \}", pl)
    assert m and (int(m.group(1)), int(m.group(2))) == (65536, 131072)
    hdr = open(os.path.join(ROOT, "include", "mldsa_hip.h")).read()
    doc = hdr[hdr.index("#define MLDSA_OPT_SIGN_LANES 7"):hdr.index("#define MLDSA_OPT_SIGN_CT0_EXACT")]
    assert "0 (default)" in doc and "131 072" in doc and "65 536" in doc
    assert "opt_sign_lanes = 0" in open(os.path.join(CSRC, "ctx.h")).read()
    assert "export_sigs ? 1 : sign_lanes_for(ctx, set, chunk)" in pl
    assert "std::max(layout(1), layout(2))" in pl
