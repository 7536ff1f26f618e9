// What the host-only parts of the library share (error reporting, device binding, the capture lock): no device code, so that
// csrc/batcher.cpp -- plain C++ -- builds with any host compiler (tests/test_cabi_cpu.py runs it under ThreadSanitizer with g++).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#include <mutex>
#include <shared_mutex>

#include "../../include/mldsa_hip.h"

namespace mldsa {

const mldsa_params *params_of(int set);
int set_error(int code, const char *what, hipError_t e = hipSuccess);

#define MLDSA_STR2(x) #x
#define MLDSA_STR(x) MLDSA_STR2(x)
// (the message names the call and where it was made: "hipGetLastError() [kernels_small.hip:412]: ...")
#define MLDSA_HIP_CHECK(expr)                                                                                              \
    do {                                                                                                                   \
        hipError_t _e = (expr);                                                                                            \
        if (_e != hipSuccess) return mldsa::set_error(MLDSA_ERR_DEVICE, #expr " [" __FILE__ ":" MLDSA_STR(__LINE__) "]", _e); \
    } while (0)

// A HIP call whose failure the caller deliberately tolerates (best-effort ordering / bookkeeping).  The failure is still worth seeing when
// something is being debugged: MLDSA_DEBUG_IGNORED=1 prints it.  The thread's sticky "last error" is cleared so that the next
// launcher's hipGetLastError() check reports ITS OWN launch, not this.
inline void tolerate(hipError_t e, const char *what) {
    if (e == hipSuccess) return;
    static const bool verbose = [] { const char *v = getenv("MLDSA_DEBUG_IGNORED"); return v && *v == '1'; }();
    if (verbose) fprintf(stderr, "mldsa_hip: tolerated %s: %s\n", what, hipGetErrorString(e));
    (void)hipGetLastError();
}
#define MLDSA_TOLERATE(expr) mldsa::tolerate((expr), #expr " [" __FILE__ ":" MLDSA_STR(__LINE__) "]")

// Every extern "C" entry that touches HIP runs under one of these: the calling thread is bound to the
// context's device for the duration of the call and its previous device is restored afterwards, so a
// context created for device N works from any host thread and next to contexts of other devices.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// Stream capture against device-wide synchronisation.  hipFree, hipHostFree and hipDeviceSynchronize wait for EVERY stream of the
// device, and waiting for a stream that another thread is capturing invalidates that capture ("operation failed due to a previous
// error during capture": found with four batcher lanes on one GPU, one growing a staging buffer while another captured a signing
// call).  Several contexts of one process -- a group's workers, a batcher's lanes -- therefore share this lock: a capture holds it
// shared from hipStreamBeginCapture to hipStreamEndCapture, every device-wide synchronisation of the library holds it exclusively.
// (The application's own hipFree cannot be seen from here: run_op falls back to launching directly when a capture does not end well.)
std::shared_mutex &capture_mutex();
struct Quiesce {  // scope of a device-wide synchronisation (not recursive: never around a call that takes it again)
    std::unique_lock<std::shared_mutex> lk;
    Quiesce() : lk(capture_mutex()) {}
};
inline hipError_t device_sync_quiesced() { Quiesce q; return hipDeviceSynchronize(); }
inline hipError_t free_quiesced(void *dev_ptr) { Quiesce q; return hipFree(dev_ptr); }
inline hipError_t host_free_quiesced(void *host_ptr) { Quiesce q; return hipHostFree(host_ptr); }
// allocations are on the runtime's list of calls that may not run beside a capture either; synchronous copies and memsets on the
// NULL stream wait for every blocking stream of the device
inline hipError_t malloc_quiesced(void **dev_ptr, size_t bytes) { Quiesce q; return hipMalloc(dev_ptr, bytes); }
inline hipError_t host_malloc_quiesced(void **host_ptr, size_t bytes, unsigned flags = 0) { Quiesce q; return hipHostMalloc(host_ptr, bytes, flags); }
inline hipError_t memcpy_quiesced(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) { Quiesce q; return hipMemcpy(dst, src, bytes, kind); }
inline hipError_t memset_quiesced(void *dev_ptr, int value, size_t bytes) { Quiesce q; return hipMemset(dev_ptr, value, bytes); }

// Host-side clearing of secret bytes (staged private keys, seeds, rnd; the reference's `zeroize`, types.rs:19, 45).  A plain memset
// of memory that is freed or never read again is a dead store the compiler may drop; this one cannot be: volatile stores, then a
// compiler barrier that names the memory.  Every host-side clearing site of the library goes through it
// (tests/test_sanitizers_cpu.py checks a freed staging buffer of the batcher for key bytes).
inline void wipe_host(void *p, size_t bytes) {
#ifdef MLDSA_TEST_NO_ZEROISE
    (void)p; (void)bytes;  // negative control of the residue tests: the probe must FIND the secrets when nothing clears them
    return;
#endif
    if (!p || !bytes) return;
    volatile unsigned char *v = static_cast<volatile unsigned char *>(p);
    size_t i = 0;
    if ((reinterpret_cast<uintptr_t>(p) & 7) == 0) {
        volatile uint64_t *w = static_cast<volatile uint64_t *>(p);
        for (; i + 8 <= bytes; i += 8) w[i >> 3] = 0;
    }
    for (; i < bytes; i++) v[i] = 0;
    __asm__ __volatile__("" : : "r"(p) : "memory");
}

}  // namespace mldsa
