// Practical HBM bandwidth on this box for the access shapes the ML-DSA kernels use:
// streaming read (sum), streaming copy, and 6:1 read:write (the fused verify-arith mix).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k_read(const int4 *in, int4 *out, size_t n) {
    int4 acc = make_int4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        int4 v = in[i];
        acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678) out[0] = acc;
}
__global__ __launch_bounds__(256) void k_copy(const int4 *in, int4 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
// read R chunks of 1 KiB-per-wave rows, write one: ratio like A_hat rows in, w out
template <int R>
__global__ __launch_bounds__(256) void k_mix(const int4 *in, int4 *out, size_t n_out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_out; i += (size_t)gridDim.x * 256) {
        int4 acc = make_int4(0, 0, 0, 0);
        const size_t row = i / 64, lane = i % 64;
#pragma unroll
        for (int r = 0; r < R; r++) {
            int4 v = in[(row * R + r) * 64 + lane];
            acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
        }
        out[i] = acc;
    }
}
int main() {
    const size_t bytes = (size_t)2 << 30;
    int4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n = bytes / 16;
    for (int bpc : {4, 8, 16, 32}) {
        const unsigned grid = 256 * bpc;
        float ms;
        auto time = [&](auto f, double moved, const char *name) {
            for (int i = 0; i < 3; i++) f();
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10; i++) f();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            printf("blocks/CU %2d  %-10s %.0f GB/s\n", bpc, name, moved / (ms / 10 * 1e-3) / 1e9);
        };
        time([&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, n); }, (double)bytes, "read");
        time([&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes, "copy");
        const size_t n_out = n / 6 / 64 * 64;
        time([&] { hipLaunchKernelGGL(k_mix<6>, dim3(grid), dim3(256), 0, 0, a, b, n_out); }, (double)n_out * 16 * 7, "mix 6:1");
    }
    return 0;
}
