// Which HIP streams share a hardware queue?  A long spin kernel on stream i, a tiny kernel on stream j: if the tiny one only
// finishes with the spin, the two streams are serialised (same hardware queue, barrier bits).  Prints the conflict matrix.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_spin(unsigned* p, long iters) {
    unsigned v = threadIdx.x;
    for (long i = 0; i < iters; i++) v = v * 1664525u + 1013904223u;
    if (v == 0x1234567u) p[0] = v;
}
__global__ void k_tiny(unsigned* p) { if (threadIdx.x == 0) p[1] = 1; }
int main() {
    const int N = 10;
    hipStream_t s[N];
    for (int i = 0; i < N; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
    unsigned* d; hipMalloc(&d, 4096);
    // calibrate the spin to ~2 ms
    long iters = 200000;
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[0], d, iters); hipDeviceSynchronize();
    double t0 = now(); hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[0], d, iters); hipDeviceSynchronize();
    double spin_ms = (now() - t0) * 1e3;
    printf("spin kernel: %.2f ms\n     ", spin_ms);
    for (int j = 0; j < N; j++) printf(" s%-2d", j);
    printf("   (X = a tiny kernel on the column's stream waits for a spin on the row's stream)\n");
    for (int i = 0; i < N; i++) {
        printf("s%-2d  ", i);
        for (int j = 0; j < N; j++) {
            if (i == j) { printf("  . "); continue; }
            hipDeviceSynchronize();
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[i], d, iters);
            double a = now();
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[j], d);
            hipStreamSynchronize(s[j]);
            double ms = (now() - a) * 1e3;
            printf("  %c ", ms > 0.5 * spin_ms ? 'X' : '-');
        }
        printf("\n");
    }
    hipDeviceSynchronize();
    // the same question for a device -> host copy: does a 64 MiB hipMemcpyAsync on the row's stream hold back a tiny kernel on
    // the column's stream?
    void *dev, *host;
    const size_t bytes = 64u << 20;
    hipMalloc(&dev, bytes); hipHostMalloc(&host, bytes, hipHostMallocDefault);
    hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s[0]); hipDeviceSynchronize();
    t0 = now(); hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s[0]); hipDeviceSynchronize();
    const double copy_ms = (now() - t0) * 1e3;
    printf("64 MiB D2H: %.2f ms\n     ", copy_ms);
    for (int j = 0; j < N; j++) printf(" s%-2d", j);
    printf("   (X = a tiny kernel on the column's stream waits for a D2H copy on the row's stream)\n");
    for (int i = 0; i < N; i++) {
        printf("s%-2d  ", i);
        for (int j = 0; j < N; j++) {
            hipDeviceSynchronize();
            hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s[i]);
            double a = now();
            hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[j], d);
            hipStreamSynchronize(s[j]);
            double ms = (now() - a) * 1e3;
            printf("  %c ", ms > 0.5 * copy_ms ? 'X' : '-');
        }
        printf("\n");
    }
    hipDeviceSynchronize();
    // Is there ONE in-order path for copies?  A D2H copy on s1 that has to wait for a spin kernel on s0 (event), then a small
    // H2D copy on s2 with no dependency at all: when does the H2D finish?
    {
        hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        void* hsmall; hipHostMalloc(&hsmall, 1 << 20, hipHostMallocDefault);
        for (int variant = 0; variant < 2; variant++) {
            hipDeviceSynchronize();
            const double a = now();
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s[0], d, iters);
            hipEventRecord(ev, s[0]);
            if (variant == 0) {  // D2H (blocked behind the spin) submitted BEFORE the independent H2D
                hipStreamWaitEvent(s[1], ev, 0);
                hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s[1]);
                hipMemcpyAsync(dev, hsmall, 1 << 20, hipMemcpyHostToDevice, s[2]);
            } else {             // the other way round
                hipMemcpyAsync(dev, hsmall, 1 << 20, hipMemcpyHostToDevice, s[2]);
                hipStreamWaitEvent(s[1], ev, 0);
                hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s[1]);
            }
            hipStreamSynchronize(s[2]);
            const double h2d_done = (now() - a) * 1e3;
            hipDeviceSynchronize();
            printf("%s: the independent 1 MiB H2D finished after %.2f ms (spin %.2f ms, then the D2H %.2f ms)\n",
                   variant == 0 ? "D2H submitted first " : "H2D submitted first ", h2d_done, spin_ms, copy_ms);
        }
    }
    return 0;
}