// Stand-ins for the few HIP runtime calls csrc/batcher.cpp makes, and for the library internals it links against (error text,
// parameter table, the capture lock), so that the REAL batcher -- its queues, spinlock, futex wake-ups, key table -- runs on the CPU
// under ThreadSanitizer (tests/test_cabi_cpu.py; the GPU pool has no sanitizers).  "Device" memory is heap memory, a stream is an
// opaque tag, every asynchronous call completes at once.  Test infrastructure only.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>

#include "../../include/mldsa_hip.h"

extern "C" {
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
void stub_free_hook(const void *p, size_t bytes) __attribute__((weak));
static std::mutex g_dev_mu;
static std::map<void *, size_t> g_dev_sizes;
hipError_t hipMalloc(void **p, size_t n) {
    *p = std::calloc(1, n ? n : 1);
    if (*p) { std::lock_guard<std::mutex> lk(g_dev_mu); g_dev_sizes[*p] = n; }
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) {
    size_t n = 0;
    if (p) { std::lock_guard<std::mutex> lk(g_dev_mu); auto it = g_dev_sizes.find(p); if (it != g_dev_sizes.end()) { n = it->second; g_dev_sizes.erase(it); } }
    if (p && n && stub_free_hook) stub_free_hook(p, n);  // "device" memory the batcher gives back: the test looks for key bytes in it
    std::free(p);
    return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(8)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
}

namespace mldsa {
static thread_local std::string g_msg;
int set_error(int code, const char *what, hipError_t) { g_msg = what ? what : ""; return code; }
const mldsa_params *params_of(int set) {
    static mldsa_params p[3];
    static const bool init = [] { return mldsa_get_params(44, &p[0]) == 0 && mldsa_get_params(65, &p[1]) == 0 && mldsa_get_params(87, &p[2]) == 0; }();
    (void)init;
    return set == 44 ? &p[0] : set == 65 ? &p[1] : set == 87 ? &p[2] : nullptr;
}
std::shared_mutex &capture_mutex() {
    static std::shared_mutex m;
    return m;
}
}  // namespace mldsa
