// Can the DMA path scatter rows?  hipMemcpyBatchAsync of N rows of 3309 bytes (device, completion order -> page-locked host, op order),
// against one contiguous hipMemcpyAsync of the same bytes: the question behind mldsa_sign_host's per-round export.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t row = 3309, n = 13000;
    uint8_t *dev, *host;
    CK(hipMalloc(&dev, n * row)); CK(hipHostMalloc(&host, 65536 * row, hipHostMallocDefault));
    CK(hipMemset(dev, 7, n * row));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::vector<size_t> perm(65536); std::iota(perm.begin(), perm.end(), 0); std::shuffle(perm.begin(), perm.end(), std::mt19937(1));
    std::vector<void*> dsts(n), srcs(n); std::vector<size_t> sizes(n, row);
    for (size_t i = 0; i < n; i++) { srcs[i] = dev + i * row; dsts[i] = host + perm[i] * row; }
    for (int rep = 0; rep < 3; rep++) {
        double t0 = now();
        CK(hipMemcpyAsync(host, dev, n * row, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
        const double tc = now() - t0;
        t0 = now();
        size_t fail = 0;
        hipMemcpyAttributes attr{}; attr.srcAccessOrder = hipMemcpySrcAccessOrderStream; size_t idx0 = 0;
        hipError_t e = hipMemcpyBatchAsync(dsts.data(), srcs.data(), sizes.data(), n, &attr, &idx0, 1, &fail, s);
        const double tsub = now() - t0;
        if (e != hipSuccess) { printf("hipMemcpyBatchAsync: %s (fail index %zu)\n", hipGetErrorString(e), fail); return 0; }
        CK(hipStreamSynchronize(s));
        const double tb = now() - t0;
        printf("contiguous %zu x %zu B: %.0f us (%.1f GB/s) | batch of %zu rows: submit %.0f us, done %.0f us (%.1f GB/s)\n", n, row, tc * 1e6,
               n * row / tc / 1e9, n, tsub * 1e6, tb * 1e6, n * row / tb / 1e9);
    }
    return 0;
}
