// How should a KERNEL write to page-locked host memory so that kernels running beside it are not slowed?  (mldsa_sign_host's
// per-round export of finished signatures, csrc/kernels_sign.hip k_export_done.)  For several shapes of a device -> host copy
// kernel -- workgroups, and whether a wave waits for its store before issuing the next -- prints the copy's own bandwidth and
// how long a 1 GiB device fill (HBM-write-bound) and a VALU-bound spin kernel take on another stream while it runs.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_fill(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(1, 2, 3, 4);
}
__global__ void k_spin(unsigned* p, int iters) {
    unsigned v = threadIdx.x + blockIdx.x;
    for (int i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    if (v == 0x2545F491u) p[0] = v;
}
template <bool WAIT>
__global__ __launch_bounds__(256) void k_down(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        dst[i] = src[i];
        if (WAIT) __builtin_amdgcn_s_waitcnt(0);
    }
}
int main() {
    const size_t bytes = 216u << 20, fill_bytes = 1u << 30;
    void *dev, *host, *scratch;
    hipMalloc(&dev, bytes); hipHostMalloc(&host, bytes, hipHostMallocDefault); hipMalloc(&scratch, fill_bytes);
    hipMemset(dev, 5, bytes);
    hipStream_t s1, s2, s3;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking); hipStreamCreateWithFlags(&s3, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto fill_ms = [&](hipStream_t s) { hipEventRecord(e0, s); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s, (uint4*)scratch, fill_bytes / 16); hipEventRecord(e1, s); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
    auto spin_ms = [&](hipStream_t s) { hipEventRecord(e0, s); hipLaunchKernelGGL(k_spin, dim3(4096), dim3(256), 0, s, (unsigned*)scratch, 20000); hipEventRecord(e1, s); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms; };
    fill_ms(s2); spin_ms(s2);
    printf("alone: fill 1 GiB %.0f us, spin %.0f us\n", fill_ms(s2) * 1e3, spin_ms(s2) * 1e3);
    for (hipStream_t other : {s2, s3})
    for (int wait = 0; wait < 2; wait++)
        for (int wgs : {2, 4, 8, 16, 32, 128}) {
            auto down = [&]() { if (wait) hipLaunchKernelGGL(k_down<true>, dim3(wgs), dim3(256), 0, s1, (const uint4*)dev, (uint4*)host, bytes / 16);
                                else hipLaunchKernelGGL(k_down<false>, dim3(wgs), dim3(256), 0, s1, (const uint4*)dev, (uint4*)host, bytes / 16); };
            down(); hipStreamSynchronize(s1);
            double t0 = now(); down(); hipStreamSynchronize(s1); const double alone = now() - t0;
            down(); const float f = fill_ms(other); hipStreamSynchronize(s1);
            down(); const float sp = spin_ms(other); hipStreamSynchronize(s1);
            printf("k_down %3d WG%s (other stream %d): %6.1f GB/s alone | beside it: fill %5.0f us, spin %5.0f us\n", wgs, wait ? " +wait" : "      ",
                   other == s2 ? 2 : 3, bytes / alone / 1e9, f * 1e3, sp * 1e3);
        }
    // the DMA path for comparison
    { hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1);
      double t0 = now(); hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); const double alone = now() - t0;
      hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s1); const float f = fill_ms(s2); hipStreamSynchronize(s1);
      hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s1); const float sp = spin_ms(s2); hipStreamSynchronize(s1);
      printf("hipMemcpyAsync D2H: %6.1f GB/s alone | beside it: fill %5.0f us, spin %5.0f us\n", bytes / alone / 1e9, f * 1e3, sp * 1e3); }
    return 0;
}
