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
