// Do independent streams overlap latency-bound kernels on this box?  Each kernel is a
// dependent integer chain of ~`iters` steps on `blocks` single-wave workgroups.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void spin(unsigned *out, int iters) {
    unsigned x = threadIdx.x + blockIdx.x * 64;
    for (int i = 0; i < iters; i++) x = x * 1664525u + 1013904223u;
    if (x == 0xdeadbeef) out[0] = x;
}
int main(int argc, char **argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 20000;
    int blocks = argc > 2 ? atoi(argv[2]) : 128;
    int per_stream = 50;
    unsigned *d;
    hipMalloc(&d, 4);
    for (int ns : {1, 2, 4, 8}) {
        std::vector<hipStream_t> st(ns);
        for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        for (int rep = 0; rep < 2; rep++) {
            hipDeviceSynchronize();
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < per_stream; k++)
                for (auto &s : st) hipLaunchKernelGGL(spin, dim3(blocks), dim3(64), 0, s, d, iters);
            hipDeviceSynchronize();
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("streams=%d kernels=%d total %.3f ms  => %.1f us per kernel-slot\n", ns, ns * per_stream, ms, ms * 1e3 / per_stream);
        }
        for (auto &s : st) hipStreamDestroy(s);
    }
    return 0;
}
