// Do a throughput-bound kernel and a latency-bound kernel of the signing round overlap when they are issued on two
// streams?  A = ExpandMask for 65 536 ML-DSA-65 slots (integer-issue-bound, ~0.23 ms); B = SampleInBall for 65 536
// slots or ExpandA for 2 048 ops (one wave per SIMD or less: latency-bound).  Prints A, B, A then B on one stream, and
// A || B on two streams (gap = what the pair costs beyond A alone).  Links against libmldsa_hip.so (seam-level ABI).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/mldsa_hip.h"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { if ((x) != 0) { fprintf(stderr, "failed: %s\n", #x); exit(1); } } while (0)
int main() {
    mldsa_ctx* ctx;
    CK(mldsa_ctx_create(0, &ctx));
    const size_t n = 65536, n_small = 2048;
    uint8_t *rho_pp, *ct, *rho;
    uint16_t* kappa;
    int32_t *y, *c, *a;
    hipMalloc(&rho_pp, n * 64); hipMalloc(&ct, n * 48); hipMalloc(&rho, n_small * 32); hipMalloc(&kappa, n * 2);
    hipMalloc(&y, n * 5 * 1024); hipMalloc(&c, n * 1024); hipMalloc(&a, n_small * 30 * 1024);
    hipMemset(rho_pp, 7, n * 64); hipMemset(ct, 9, n * 48); hipMemset(rho, 3, n_small * 32); hipMemset(kappa, 0, n * 2);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    hipEvent_t e1, e2;
    hipEventCreateWithFlags(&e1, hipEventDisableTiming); hipEventCreateWithFlags(&e2, hipEventDisableTiming);
    auto A = [&](hipStream_t s) { CK(mldsa_expand_mask(ctx, MLDSA_65, rho_pp, kappa, y, n, s)); };
    auto B1 = [&](hipStream_t s) { CK(mldsa_sample_in_ball(ctx, MLDSA_65, ct, c, n, s)); };
    auto B2 = [&](hipStream_t s) { CK(mldsa_expand_a(ctx, MLDSA_65, rho, a, n_small, s)); };
    const int reps = 200;
    auto run = [&](const char* name, auto f) {
        for (int i = 0; i < 10; i++) f();
        hipDeviceSynchronize();
        const double t0 = now();
        for (int i = 0; i < reps; i++) f();
        hipDeviceSynchronize();
        printf("%-44s %7.1f us\n", name, (now() - t0) / reps * 1e6);
    };
    // a pair on two streams, both streams joined after every pair
    auto pair = [&](auto fa, auto fb) {
        return [&, fa, fb]() {
            fa(s1); fb(s2);
            hipEventRecord(e1, s1); hipEventRecord(e2, s2);
            hipStreamWaitEvent(s1, e2, 0); hipStreamWaitEvent(s2, e1, 0);
        };
    };
    run("A  ExpandMask 65536", [&]() { A(s1); });
    run("B1 SampleInBall 65536", [&]() { B1(s1); });
    run("B2 ExpandA 2048", [&]() { B2(s1); });
    run("A ; B1 one stream", [&]() { A(s1); B1(s1); });
    run("A || B1 two streams", pair(A, B1));
    run("B1 || A two streams (B first)", [&]() { B1(s2); A(s1); hipEventRecord(e1, s1); hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0); hipStreamWaitEvent(s2, e1, 0); });
    run("A ; B2 one stream", [&]() { A(s1); B2(s1); });
    run("A || B2 two streams", pair(A, B2));
    run("B2 || A two streams (B first)", [&]() { B2(s2); A(s1); hipEventRecord(e1, s1); hipEventRecord(e2, s2); hipStreamWaitEvent(s1, e2, 0); hipStreamWaitEvent(s2, e1, 0); });
    mldsa_ctx_destroy(ctx);
    return 0;
}
