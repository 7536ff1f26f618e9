// A stand-in for the HIP runtime under the WHOLE library's host code (capi / pipeline / host_api / group .hip, the launchers of the
// kernel files, batcher.cpp), so that those ~3 400 threaded host lines run on the CPU under ASan + UBSan and under ThreadSanitizer
// (tests/test_sanitizers_cpu.py::test_host_orchestration_under_sanitizers; the GPU pool has no sanitizers).  The sources are compiled
// with `hipcc --offload-host-only`: kernels become host stubs that arrive in hipLaunchKernel below.
//
// Model: "device" memory is heap memory of EXACTLY the requested size (ASan sees every host-side overrun of a staging copy or a
// carve); streams, events and graphs are opaque tags and everything completes at once; `n_devices` devices exist.  A kernel launch is
// a no-op -- round counters stay "all finished", so the host's end-of-call checks pass -- EXCEPT for a touch model of the kernels
// whose byte ranges are pure host arithmetic: k_zero / k_zero_if_done (the clearing spans of the workspace), k_copy_rows, k_mu (reads
// the message and ctx bytes its offset tables name, writes mu and the refusal flags), the fixed-shape hash with its verdict epilogue
// (ok[op]), k_init_active (status and signature rows), k_sanitize_keys, the key-generation epilogue (pk / sk rows), k_count_nonzero and
// the kernarg self-test.  A span that leaves its allocation, a stale workspace pointer or a mis-sized pass shows up as an ASan report.
// hipMalloc fails above a settable limit (stub_set_mem_limit): the workspace-shrinking paths run.  Test infrastructure only.
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mldsa_hip.h"

namespace {
// (constructed on first use and never destroyed: the kernel registration constructors of the library's objects run before this file's
//  own static initialisers, and the library's static destructors may still free memory after them)
struct State {
    std::mutex mu;
    std::map<const void *, std::string> kernels;          // host stub address -> device (mangled) name
    std::map<void *, size_t> dev, pinned;                 // live allocations
    std::map<hipStream_t, bool> capturing;
};
State &S() { static State *s = new State; return *s; }
#define g_mu S().mu
#define g_kernels S().kernels
#define g_dev S().dev
#define g_pinned S().pinned
#define g_capturing S().capturing
std::atomic<size_t> g_mem_limit{(size_t)1 << 40}, g_launches{0}, g_touches{0}, g_allocs{0}, g_failed_allocs{0};
std::atomic<int> g_n_devices{8};
thread_local int t_device = 0;
thread_local bool t_exempt = false;  // the test driver's own buffers are not subject to the memory limit
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local std::vector<CallCfg> t_cfg;

template <class T> T arg(void **args, int i) { T v; std::memcpy(&v, args[i], sizeof(T)); return v; }
// volatile reads / writes of a byte range: the compiler keeps them, ASan / TSan see them
void touch_read(const void *p, size_t n) {
    const volatile uint8_t *b = static_cast<const volatile uint8_t *>(p);
    uint8_t acc = 0;
    for (size_t i = 0; i < n; i++) acc ^= b[i];
    (void)acc;
}
int template_int(const std::string &name, int index) {  // the index-th "Li<N>E" / "Lb<N>E" of a mangled template argument list
    size_t pos = name.find('I');
    for (int k = 0; pos != std::string::npos; k++) {
        pos = name.find_first_of("L", pos);
        if (pos == std::string::npos) break;
        size_t e = name.find('E', pos);
        if (k == index) return std::atoi(name.c_str() + pos + 2);
        pos = e;
    }
    return -1;
}

// (copies of the by-value argument structs that are defined in .hip files; tests/test_sanitizers_cpu.py compares them with the sources)
struct VerdictArgs { const uint8_t *sigs; size_t sig_len; const int32_t *znorm; int32_t zbound; const int32_t *hvalid; const int32_t *ctx_bad; uint8_t *ok; };
struct KeygenOut { uint8_t *pk, *sk; size_t pk_len, sk_len, t0_off; int eta, ebits; };

void emulate(const std::string &k, void **a) {
    auto has = [&](const char *s) { return k.find(s) != std::string::npos; };
    if (has("6k_zeroE")) {
        std::memset(arg<uint8_t *>(a, 0), 0, arg<size_t>(a, 1));
    } else if (has("14k_zero_if_doneE")) {
        const uint32_t *ctl = arg<const uint32_t *>(a, 0);
        if (ctl[arg<int>(a, 1)] == 0) std::memset(arg<uint8_t *>(a, 2), 0, arg<size_t>(a, 3));
    } else if (has("11k_copy_rowsE")) {
        uint8_t *dst = arg<uint8_t *>(a, 0); const size_t ds = arg<size_t>(a, 1);
        const uint8_t *src = arg<const uint8_t *>(a, 2); const size_t ss = arg<size_t>(a, 3);
        const int rb = arg<int>(a, 4); const size_t n = arg<size_t>(a, 5);
        for (size_t r = 0; r < n; r++) std::memcpy(dst + r * ds, src + r * ss, (size_t)(rb / 4) * 4);
    } else if (has("19k_late_arg_selftestE")) {
        *arg<uint32_t *>(a, 1) = 0x80000000u;  // "the kernel ran, no field differs"
    } else if (has("15k_sanitize_keysE")) {
        const uint32_t *ki = arg<const uint32_t *>(a, 0); const uint32_t nk = arg<uint32_t>(a, 1); const size_t n = arg<size_t>(a, 2);
        uint32_t *safe = arg<uint32_t *>(a, 3); int32_t *bad = arg<int32_t *>(a, 4);
        for (size_t i = 0; i < n; i++) { const uint32_t v = ki[i]; safe[i] = v < nk ? v : 0u; bad[i] = v < nk ? 0 : 2; }
    } else if (has("15k_count_nonzeroE")) {
        const uint8_t *src = arg<const uint8_t *>(a, 0); const size_t n = arg<size_t>(a, 1); unsigned long long c = 0;
        for (size_t i = 0; i < n; i++) c += src[i] != 0;
        *arg<unsigned long long *>(a, 2) += c;
    } else if (has("13k_init_activeE")) {
        // (n, bad_op, done, kappa, status, act_out, ctl, sigs, sig_len); modelled as "every op is refused": the rows a refused op touches
        const size_t n = arg<size_t>(a, 0); const int32_t *bad = arg<const int32_t *>(a, 1); int32_t *done = arg<int32_t *>(a, 2);
        uint16_t *kappa = arg<uint16_t *>(a, 3); int32_t *status = arg<int32_t *>(a, 4); uint8_t *sigs = arg<uint8_t *>(a, 7);
        const size_t sl = arg<size_t>(a, 8);
        for (size_t i = 0; i < n; i++) {
            kappa[i] = 0; done[i] = 1 | bad[i];
            if (status) status[i] = MLDSA_OK;
            std::memset(sigs + i * sl, 0, sl);
        }
    } else if (has("4k_muE") || has("9k_mu_coopE")) {
        // (tr, tr_stride, key_idx, mode, msgs, msg_off, ctxs, ctx_off, mu, mu_stride, ctx_bad, key_bad, n_ops, op0, n_call)
        const uint8_t *tr = arg<const uint8_t *>(a, 0); const size_t ts = arg<size_t>(a, 1); const uint32_t *ki = arg<const uint32_t *>(a, 2);
        const uint8_t *msgs = arg<const uint8_t *>(a, 4); const uint64_t *mo = arg<const uint64_t *>(a, 5);
        const uint8_t *ctxs = arg<const uint8_t *>(a, 6); const uint64_t *co = arg<const uint64_t *>(a, 7);
        uint8_t *mu = arg<uint8_t *>(a, 8); const size_t ms = arg<size_t>(a, 9); int32_t *ctx_bad = arg<int32_t *>(a, 10);
        const int32_t *key_bad = arg<const int32_t *>(a, 11); const size_t n = arg<size_t>(a, 12), op0 = arg<size_t>(a, 13), nc = arg<size_t>(a, 14);
        for (size_t op = 0; op < n; op++) {
            const uint64_t m0 = mo[op0 + op], m1 = mo[op0 + op + 1];
            bool bad_off = !(mo[0] <= m0 && m0 <= m1 && m1 <= mo[nc]);
            uint64_t c0 = 0, c1 = 0;
            bad_off |= (m1 - m0) != 0 && msgs == nullptr;
            if (co) { c0 = co[op0 + op]; c1 = co[op0 + op + 1]; bad_off |= !(co[0] <= c0 && c0 <= c1 && c1 <= co[nc]); bad_off |= (c1 - c0) != 0 && ctxs == nullptr; }
            const int flag = bad_off ? 2 : (c1 - c0) > 255 ? 1 : (key_bad ? key_bad[op] : 0);
            if (ctx_bad) ctx_bad[op] = flag;
            if (!bad_off && (c1 - c0) <= 255) {  // the op is hashed: exactly the bytes its table entries name
                touch_read(tr + (ki ? ki[op] : op) * ts, 64);
                if (m1 > m0) touch_read(msgs + m0, (size_t)(m1 - m0));
                if (c1 > c0) touch_read(ctxs + c0, (size_t)(c1 - c0));
            }
            std::memset(mu + op * ms, 0x5A, 64);
        }
    } else if (has("12k_shake256_2I") || has("17k_shake256_2_coopI")) {
        // (a, sa, la, a_idx, b, sb, lb, tail, tail_len, out, so, n_ops, n_dev, vd, b_idx), template <OUT, ...>
        const int OUT = template_int(k, 0);
        const uint8_t *pa = arg<const uint8_t *>(a, 0); const size_t sa = arg<size_t>(a, 1); const int la = arg<int>(a, 2);
        const uint32_t *ai = arg<const uint32_t *>(a, 3); const uint8_t *pb = arg<const uint8_t *>(a, 4); const size_t sb = arg<size_t>(a, 5);
        const int lb = arg<int>(a, 6); uint8_t *out = arg<uint8_t *>(a, 9); const size_t so = arg<size_t>(a, 10); size_t n = arg<size_t>(a, 11);
        const uint32_t *n_dev = arg<const uint32_t *>(a, 12); const VerdictArgs vd = arg<VerdictArgs>(a, 13); const uint32_t *bi = arg<const uint32_t *>(a, 14);
        if (n_dev) n = *n_dev;
        for (size_t op = 0; op < n; op++) {
            touch_read(pa + (ai ? ai[op] : op) * sa, (size_t)la);
            if (pb) touch_read(pb + (bi ? bi[op] : op) * sb, (size_t)lb);
            if (vd.ok) {
                touch_read(vd.sigs + op * vd.sig_len, (size_t)OUT);
                vd.ok[op] = (uint8_t)(vd.znorm[op] < vd.zbound && vd.hvalid[op] && !vd.ctx_bad[op] ? 0 : 0);
            } else if (out) {
                std::memset(out + op * so, 0xA5, (size_t)OUT);
            }
        }
    } else if (has("14k_verify_arithI")) {
        // <K, L, HAS_C, W1, APACK, KG, YGB>(a_hat, a_idx, z, c, t1, key_idx, w_out, n_ops, ..., n_dev [15], z_idx [16], kg [17], yr [18])
        const int K = template_int(k, 0); const bool has_c = template_int(k, 2) != 0, kg_form = template_int(k, 5) != 0;
        size_t n = arg<size_t>(a, 7); const uint32_t *n_dev = arg<const uint32_t *>(a, 15);
        if (n_dev) n = *n_dev;
        if (kg_form) {
            const KeygenOut kg = arg<KeygenOut>(a, 17);
            for (size_t op = 0; op < n; op++) { std::memset(kg.pk + op * kg.pk_len + 32, 0x11, kg.pk_len - 32); std::memset(kg.sk + op * kg.sk_len + 128, 0x22, kg.sk_len - 128); }
        } else if (has_c) {
            int32_t *w = arg<int32_t *>(a, 6);
            for (size_t op = 0; op < n; op++) std::memset(w + op * (size_t)K * 256, 0, (size_t)K * 1024);
        }
    } else {
        return;
    }
    g_touches++;
}
}  // namespace

extern "C" {
// ---- test hooks
void stub_set_mem_limit(size_t bytes) { g_mem_limit = bytes; }
void stub_exempt_from_limit(int on) { t_exempt = on != 0; }
void stub_set_devices(int n) { g_n_devices = n; }
size_t stub_launches(void) { return g_launches; }
size_t stub_touches(void) { return g_touches; }
size_t stub_failed_allocs(void) { return g_failed_allocs; }
size_t stub_live_allocations(void) { std::lock_guard<std::mutex> lk(g_mu); return g_dev.size() + g_pinned.size(); }

// ---- registration and launch
void **__hipRegisterFatBinary(const void *) { static void *dummy[4]; return dummy; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *host_fn, char *, const char *device_name, unsigned, void *, void *, void *, void *, int *) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_kernels[host_fn] = device_name ? device_name : "";
}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    t_cfg.push_back({grid, block, shmem, stream});
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream) {
    if (t_cfg.empty()) return hipErrorInvalidValue;
    const CallCfg c = t_cfg.back();
    t_cfg.pop_back();
    *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *fn, dim3 grid, dim3 block, void **args, size_t, hipStream_t) {
    g_launches++;
    if (grid.x == 0 || block.x == 0 || block.x * block.y * block.z > 1024) { std::fprintf(stderr, "stub: invalid launch configuration\n"); std::abort(); }
    std::string name;
    { std::lock_guard<std::mutex> lk(g_mu); auto it = g_kernels.find(fn); if (it != g_kernels.end()) name = it->second; }
    if (!name.empty()) emulate(name, args);
    return hipSuccess;
}

// ---- devices
hipError_t hipGetDeviceCount(int *n) { *n = g_n_devices; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= g_n_devices) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int) { std::memset(p, 0, sizeof(*p)); p->multiProcessorCount = 256; p->totalGlobalMem = (size_t)288 << 30; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : "stub error"; }

// ---- memory
hipError_t hipMalloc(void **p, size_t n) {
    *p = nullptr;
    if (n > g_mem_limit && !t_exempt) { g_failed_allocs++; return hipErrorOutOfMemory; }
    *p = std::calloc(1, n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    g_allocs++;
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[*p] = n;
    return hipSuccess;
}
hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    { std::lock_guard<std::mutex> lk(g_mu); if (!g_dev.erase(p)) { std::fprintf(stderr, "stub: hipFree of a pointer hipMalloc did not return\n"); std::abort(); } }
    std::free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    *p = std::calloc(1, n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> lk(g_mu);
    g_pinned[*p] = n;
    return hipSuccess;
}
hipError_t hipHostFree(void *p) {
    if (!p) return hipSuccess;
    { std::lock_guard<std::mutex> lk(g_mu); if (!g_pinned.erase(p)) { std::fprintf(stderr, "stub: hipHostFree of a pointer hipHostMalloc did not return\n"); std::abort(); } }
    std::free(p);
    return hipSuccess;
}
static bool find_pinned(const void *p, void **base, size_t *size) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_pinned.upper_bound(const_cast<void *>(p));
    if (it == g_pinned.begin()) return false;
    --it;
    if (static_cast<const char *>(p) >= static_cast<const char *>(it->first) + it->second) return false;
    *base = it->first; *size = it->second;
    return true;
}
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned) { void *b; size_t s; if (!find_pinned(host, &b, &s)) return hipErrorInvalidValue; *dev = host; return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *p) {
    void *b; size_t s;
    if (!find_pinned(p, &b, &s)) return hipErrorInvalidValue;  // pageable memory: what the real runtime says
    std::memset(a, 0, sizeof(*a));
    a->type = hipMemoryTypeHost; a->hostPointer = const_cast<void *>(p); a->devicePointer = const_cast<void *>(p);
    return hipSuccess;
}
hipError_t hipMemGetAddressRange(hipDeviceptr_t *base, size_t *size, hipDeviceptr_t p) {
    void *b; size_t s;
    if (!find_pinned(p, &b, &s)) return hipErrorInvalidValue;
    *base = b; *size = s;
    return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }

// ---- streams, events, graphs: opaque tags, everything has completed
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(16)); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { { std::lock_guard<std::mutex> lk(g_mu); g_capturing.erase(s); } std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t e, unsigned) { return e ? hipSuccess : hipErrorInvalidHandle; }
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus *st) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_capturing.find(s);
    *st = it != g_capturing.end() && it->second ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone;
    return hipSuccess;
}
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode) { std::lock_guard<std::mutex> lk(g_mu); g_capturing[s] = true; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t *g) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_capturing[s] = false;
    *g = reinterpret_cast<hipGraph_t>(std::malloc(16));
    return hipSuccess;
}
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, hipGraphNode_t *, char *, size_t) { if (!g) return hipErrorInvalidValue; *e = reinterpret_cast<hipGraphExec_t>(std::malloc(16)); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { touch_read(e, 16); return hipSuccess; }  // (a destroyed exec: ASan)
hipError_t hipGraphExecDestroy(hipGraphExec_t e) { std::free(e); return hipSuccess; }
hipError_t hipGraphDestroy(hipGraph_t g) { std::free(g); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = reinterpret_cast<hipEvent_t>(std::malloc(16)); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { return hipEventCreateWithFlags(e, 0); }
hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { touch_read(e, 16); return hipSuccess; }  // (a destroyed event: ASan)
hipError_t hipEventSynchronize(hipEvent_t e) { touch_read(e, 16); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { touch_read(e, 16); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.01f; return hipSuccess; }
}
