// Launch-overhead microbenchmark: N short kernels per iteration issued (a) one by one on a stream,
// (b) as one captured hipGraph.  Prints host time per iteration and wall time per iteration.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void k_touch(unsigned* p, int spin) {
    unsigned v = p[threadIdx.x & 63];
    for (int i = 0; i < spin; i++) v = v * 1664525u + 1013904223u;
    if (v == 0x12345u) p[0] = v;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int n_k = argc > 1 ? atoi(argv[1]) : 90, iters = 200;
    unsigned* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int spin : {0, 2000, 20000}) {
        for (int blocks : {1, 1024}) {
            auto enqueue = [&]() { for (int i = 0; i < n_k; i++) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, s, d, spin); };
            for (int w = 0; w < 3; w++) enqueue();
            hipStreamSynchronize(s);
            double t0 = now(), host = 0;
            for (int it = 0; it < iters; it++) { double a = now(); enqueue(); host += now() - a; }
            hipStreamSynchronize(s);
            const double wall_direct = (now() - t0) / iters, host_direct = host / iters;
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
            enqueue();
            hipStreamEndCapture(s, &g);
            double ti = now();
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            const double inst = now() - ti;
            for (int w = 0; w < 3; w++) hipGraphLaunch(ge, s);
            hipStreamSynchronize(s);
            t0 = now(); host = 0;
            for (int it = 0; it < iters; it++) { double a = now(); hipGraphLaunch(ge, s); host += now() - a; }
            hipStreamSynchronize(s);
            const double wall_graph = (now() - t0) / iters, host_graph = host / iters;
            printf("%3d kernels x %4d blocks, spin %5d: direct host %7.1f us wall %7.1f us | graph host %7.1f us wall %7.1f us | instantiate %7.1f us\n",
                   n_k, blocks, spin, host_direct * 1e6, wall_direct * 1e6, host_graph * 1e6, wall_graph * 1e6, inst * 1e6);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
