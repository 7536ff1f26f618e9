// Does a large device -> host copy overlap with a store-heavy kernel?  (a) hipMemcpyAsync (a blit kernel on this platform),
// (b) hsa_amd_memory_async_copy (the DMA engines, if the platform lets D2H use them).  Each alone, then beside a memset
// kernel on another stream.
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_fill(uint4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = make_uint4(1, 2, 3, 4);
}
__global__ void k_tiny(unsigned* p) { if (threadIdx.x == 0) p[blockIdx.x] = 1; }
// device -> host copy kernels of our own: 32 workgroups, plain stores / non-temporal stores
template <bool NT>
__global__ __launch_bounds__(256) void k_down(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const uint4 v = src[i];
        if (NT) {
            __builtin_nontemporal_store(v.x, &dst[i].x); __builtin_nontemporal_store(v.y, &dst[i].y);
            __builtin_nontemporal_store(v.z, &dst[i].z); __builtin_nontemporal_store(v.w, &dst[i].w);
        } else dst[i] = v;
    }
}
static hsa_agent_t g_cpu, g_gpu;
static hsa_status_t on_agent(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_CPU && g_cpu.handle == 0) g_cpu = a;
    if (t == HSA_DEVICE_TYPE_GPU && g_gpu.handle == 0) g_gpu = a;
    return HSA_STATUS_SUCCESS;
}
int main() {
    const size_t bytes = 108u << 20, fill_bytes = 1u << 30;
    void *dev, *host, *scratch;
    hipMalloc(&dev, bytes); hipHostMalloc(&host, bytes, hipHostMallocDefault); hipMalloc(&scratch, fill_bytes);
    hipMemset(dev, 5, bytes);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    auto fill = [&]() { hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s2, (uint4*)scratch, fill_bytes / 16); };
    auto t = [&](const char* name, auto f, int reps = 10) {
        f(); hipDeviceSynchronize();
        const double t0 = now();
        for (int i = 0; i < reps; i++) f();
        hipDeviceSynchronize();
        printf("%-52s %8.1f us\n", name, (now() - t0) / reps * 1e6);
    };
    t("fill 1 GiB (kernel)", [&]() { fill(); hipStreamSynchronize(s2); });
    t("hipMemcpyAsync D2H 108 MiB", [&]() { hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s1); hipStreamSynchronize(s1); });
    t("hipMemcpyAsync D2H || fill", [&]() { hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s1); fill(); hipStreamSynchronize(s1); hipStreamSynchronize(s2); });
    if (hsa_init() != HSA_STATUS_SUCCESS) { printf("hsa_init failed\n"); return 0; }
    hsa_iterate_agents(on_agent, nullptr);
    hsa_signal_t sig;
    hsa_signal_create(1, 0, nullptr, &sig);
    auto hsa_copy = [&]() -> bool {
        hsa_signal_store_relaxed(sig, 1);
        hsa_status_t st = hsa_amd_memory_async_copy(host, g_cpu, dev, g_gpu, bytes, 0, nullptr, sig);
        if (st != HSA_STATUS_SUCCESS) { printf("hsa_amd_memory_async_copy failed: %d\n", (int)st); return false; }
        return true;
    };
    auto hsa_wait = [&]() { while (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED) >= 1) {} };
    if (!hsa_copy()) return 0;
    hsa_wait();
    t("hsa_amd_memory_async_copy D2H 108 MiB", [&]() { hsa_copy(); hsa_wait(); });
    t("hsa copy D2H || fill", [&]() { hsa_copy(); fill(); hsa_wait(); hipStreamSynchronize(s2); });
    // how long does the fill itself take while a copy is in flight?  (events on its stream; the copy is started first)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto fill_time = [&](const char* name, auto start_copy, auto wait_copy) {
        float sum = 0;
        for (int i = 0; i < 5; i++) {
            start_copy();
            hipEventRecord(e0, s2); fill(); hipEventRecord(e1, s2);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); sum += ms;
            wait_copy();
        }
        printf("%-52s %8.1f us\n", name, sum / 5 * 1e3);
    };
    // the signing pattern: a chain of 100 tiny kernels on s2 while a copy is in flight on s1
    auto chain_time = [&](const char* name, auto start_copy, auto wait_copy) {
        float sum = 0;
        for (int i = 0; i < 5; i++) {
            start_copy();
            hipEventRecord(e0, s2);
            for (int k = 0; k < 100; k++) hipLaunchKernelGGL(k_tiny, dim3(64), dim3(64), 0, s2, (unsigned*)scratch);
            hipEventRecord(e1, s2);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); sum += ms;
            wait_copy();
        }
        printf("%-52s %8.1f us\n", name, sum / 5 * 1e3);
    };
    void* hview = host;
    chain_time("100 tiny kernels alone", [] {}, [] {});
    chain_time("100 tiny kernels while hipMemcpyAsync D2H runs", [&]() { hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s1); }, [&]() { hipStreamSynchronize(s1); });
    chain_time("100 tiny kernels while the hsa copy runs", [&]() { hsa_copy(); }, [&]() { hsa_wait(); });
    chain_time("100 tiny kernels while k_down (plain stores) runs", [&]() { hipLaunchKernelGGL(k_down<false>, dim3(32), dim3(256), 0, s1, (const uint4*)dev, (uint4*)hview, bytes / 16); }, [&]() { hipStreamSynchronize(s1); });
    chain_time("100 tiny kernels while k_down (nontemporal) runs", [&]() { hipLaunchKernelGGL(k_down<true>, dim3(32), dim3(256), 0, s1, (const uint4*)dev, (uint4*)hview, bytes / 16); }, [&]() { hipStreamSynchronize(s1); });
    t("k_down plain 108 MiB", [&]() { hipLaunchKernelGGL(k_down<false>, dim3(32), dim3(256), 0, s1, (const uint4*)dev, (uint4*)hview, bytes / 16); hipStreamSynchronize(s1); });
    t("k_down nontemporal 108 MiB", [&]() { hipLaunchKernelGGL(k_down<true>, dim3(32), dim3(256), 0, s1, (const uint4*)dev, (uint4*)hview, bytes / 16); hipStreamSynchronize(s1); });
    fill_time("fill 1 GiB alone (events)", [] {}, [] {});
    fill_time("fill 1 GiB while hipMemcpyAsync D2H runs", [&]() { hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s1); }, [&]() { hipStreamSynchronize(s1); });
    fill_time("fill 1 GiB while the hsa copy runs", [&]() { hsa_copy(); }, [&]() { hsa_wait(); });
    printf("done\n");
    return 0;
}
