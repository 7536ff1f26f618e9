// Micro-benchmark: issue rate of the integer VALU / cross-lane instructions the ML-DSA
// kernels are built from (32-bit and 24-bit multiplies, logic ops, rotates, permlane/DPP).
// The local guides do not list integer-multiply rates for gfx950, so they are measured.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#include "../fips204_amd/csrc/keccak.h"

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ILP = 8;

__device__ unsigned long long g_clk[2];  // sum of s_memtime deltas, sum of s_memrealtime deltas (block 0, wave 0)

#define CLK_BEGIN unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#define CLK_END if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = __builtin_amdgcn_s_memtime() - t0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }

#define DEFINE_KERNEL(NAME, ASM)                                                        \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters) {             \
        CLK_BEGIN                                                                       \
        uint32_t a[ILP];                                                                \
        uint32_t b = threadIdx.x * 2654435761u + 12345u, c = blockIdx.x * 40503u + 7u;  \
        for (int j = 0; j < ILP; j++) a[j] = threadIdx.x + j * 977u + 1u;               \
        for (int i = 0; i < iters; i++) {                                               \
            _Pragma("unroll") for (int j = 0; j < ILP; j++)                             \
                asm volatile(ASM : "+v"(a[j]) : "v"(b), "v"(c));                        \
        }                                                                               \
        uint32_t s = 0;                                                                 \
        for (int j = 0; j < ILP; j++) s += a[j];                                        \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                        \
        CLK_END                                                                         \
    }

DEFINE_KERNEL(k_add, "v_add_u32 %0, %0, %1")
DEFINE_KERNEL(k_xor, "v_xor_b32 %0, %0, %1")
DEFINE_KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2")
DEFINE_KERNEL(k_cnd3, "v_cndmask_b32 %0, %0, %1, s[10:11]")
DEFINE_KERNEL(k_bfi, "v_bfi_b32 %0, %1, %0, %2")
DEFINE_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
DEFINE_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
DEFINE_KERNEL(k_mul_hi_u, "v_mul_hi_u32 %0, %0, %1")
DEFINE_KERNEL(k_mul_hi_i, "v_mul_hi_i32 %0, %0, %1")
DEFINE_KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1")
DEFINE_KERNEL(k_mul_hi_u24, "v_mul_hi_u32_u24 %0, %0, %1")
DEFINE_KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2")
DEFINE_KERNEL(k_mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEFINE_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
DEFINE_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
DEFINE_KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEFINE_KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(k_dpp_add, "v_add_u32_dpp %0, %1, %0 row_ror:8 row_mask:0xf bank_mask:0xf")
DEFINE_KERNEL(k_permlane32, "s_nop 1\n\tv_permlane32_swap_b32 %0, %1")
DEFINE_KERNEL(k_bpermute, "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)")
DEFINE_KERNEL(k_swizzle, "ds_swizzle_b32 %0, %0 offset:swizzle(BITMASK_PERM,\"0000p\")\n\ts_waitcnt lgkmcnt(0)")

// 64-bit multiply-add (one instruction produces a 64-bit result)
__global__ __launch_bounds__(256) void k_mad_u64(uint32_t* out, int iters) {
    CLK_BEGIN
    uint64_t a[ILP];
    uint32_t b = threadIdx.x * 2654435761u + 12345u;
    for (int j = 0; j < ILP; j++) a[j] = threadIdx.x + j * 977u + 1u;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < ILP; j++) {
            uint32_t lo = (uint32_t)a[j];
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[j]) : "v"(lo), "v"(b) : "vcc");
        }
    }
    uint32_t s = 0;
    for (int j = 0; j < ILP; j++) s += (uint32_t)a[j] + (uint32_t)(a[j] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    CLK_END
}

typedef void (*kern_t)(uint32_t*, int);

// Keccak-f[1600] alone, one state per lane exactly as the samplers run it: the integer-ALU ceiling of
// every SHAKE-driven kernel (ExpandA, ExpandMask, SampleInBall, the hashes).
__global__ __launch_bounds__(256) void k_keccak(uint32_t* out, int perms) {
    mldsa::KeccakState st;
    for (int i = 0; i < 25; i++) { st.lo[i] = threadIdx.x * 31u + i; st.hi[i] = blockIdx.x * 17u + i; }
    for (int p = 0; p < perms; p++) mldsa::keccak_f1600(st);
    uint32_t s = 0;
    for (int i = 0; i < 25; i++) s ^= st.lo[i] ^ st.hi[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int n_cu = prop.multiProcessorCount;
    const int blocks = n_cu * 8, iters = 16384;  // ~2 ms per kernel so the clock reading settles
    uint32_t* out;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    struct { const char* name; kern_t k; } tests[] = {
        {"v_add_u32", k_add}, {"v_xor_b32", k_xor}, {"v_bfi_b32", k_bfi}, {"v_fma_f32", k_fma}, {"v_cndmask_b32(sgpr mask)", k_cnd3},
        {"v_alignbit_b32", k_alignbit}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi_u},
        {"v_mul_hi_i32", k_mul_hi_i}, {"v_mul_u32_u24", k_mul_u24}, {"v_mul_hi_u32_u24", k_mul_hi_u24},
        {"v_mad_u32_u24", k_mad_u24}, {"v_mad_i32_i24", k_mad_i24}, {"v_lshl_add_u32", k_lshl_add},
        {"v_and_or_b32", k_and_or}, {"v_cndmask_b32", k_cndmask}, {"v_mad_u64_u32", k_mad_u64},
        {"v_mov_b32_dpp", k_dpp_mov}, {"v_add_u32_dpp", k_dpp_add}, {"v_permlane32_swap(+s_nop1)", k_permlane32},
        {"ds_bpermute_b32(+wait)", k_bpermute}, {"ds_swizzle_b32(+wait)", k_swizzle},
    };
    printf("device: %s, %d CUs, clock %d MHz\n", prop.name, n_cu, prop.clockRate / 1000);
    printf("%-28s %12s %16s %18s %10s %14s\n", "instruction", "ms", "Glane-ops/s", "cyc/instr@2.4GHz", "clock MHz", "cyc/instr@clk");
    for (auto& t : tests) {
        hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 64);  // warm-up
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, iters);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        double wave_instrs_per_simd = (double)blocks * 4 /*waves per block*/ * iters * ILP / (n_cu * 4.0);
        double cyc = best * 1e-3 * 2.4e9 / wave_instrs_per_simd;
        double glops = (double)blocks * 256 * iters * ILP / (best * 1e-3) / 1e9;
        unsigned long long clk[2] = {0, 0};
        CHECK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
        // in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
        double mhz = clk[1] ? (double)clk[0] / (double)clk[1] * 100.0 : 0.0;
        printf("%-28s %12.3f %16.1f %18.2f %10.0f %14.2f\n", t.name, best, glops, cyc, mhz, cyc * mhz / 2400.0);
    }
    for (int bpc : {1, 2, 4, 8}) {  // 1, 2, 4, 8 waves per SIMD
        const int kb = n_cu * bpc, perms = 256;
        hipLaunchKernelGGL(k_keccak, dim3(kb), dim3(256), 0, 0, out, 4);
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_keccak, dim3(kb), dim3(256), 0, 0, out, perms);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("keccak_f1600 lane-per-state, %d waves/SIMD: %.3f ms, %.2f G permutations/s, %.2f us per permutation per wave\n", bpc, best,
               (double)kb * 256 * perms / (best * 1e-3) / 1e9, best * 1e3 / perms);
    }
    return 0;
}
