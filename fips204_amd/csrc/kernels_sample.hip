// SHAKE-driven samplers for gfx950: ExpandA / RejNTTPoly, ExpandS / RejBoundedPoly,
// ExpandMask, SampleInBall.  Replaces src/hashing.rs:43-313 of the reference (and the leaf
// functions coeff_from_three_bytes / coeff_from_half_byte / bit_unpack of conversion.rs).
//
// One XOF stream per lane (keccak.h), 64 streams per wavefront.  Every lane parses its own
// squeezed block into a small lane-private LDS staging row; the wave then flushes the rows
// cooperatively so that HBM only sees contiguous runs (one coalesced store instruction per
// stream and flush) instead of 64 lanes scattering single dwords 1 KiB apart.
#include <algorithm>
#include "ctx.h"
#include "challenge_dev.h"
#include "sampler_dev.h"
#include "keccak_coop.h"
#include "expand_coop_dev.h"

namespace mldsa {

// ------------------------------------------------------------------------------------
// ExpandA (hashing.rs:225-239) = K*L x RejNTTPoly (hashing.rs:111-146):
// stream (op, r, s): SHAKE128(rho || s || r); 3 bytes -> 23-bit candidate, keep if < q
// (coeff_from_three_bytes, conversion.rs:40-61).  Output A_hat[op][r][s], canonical [0, q).
// At most FOUR waves per SIMD (the packed form needs no LDS and 72 VGPRs and would fit seven): the kernel is issue-bound from four
// on, and the verifier's mu and SampleInBall run on a second stream underneath it -- with seven ExpandA waves on every SIMD those
// one-wave-per-SIMD kernels got an eighth of the issue slots, took as long as ExpandA itself (1.2 ms instead of 0.2) and delayed
// k_verify_main by 75 us.
template <int K, int L, bool PACK24>
__global__ __launch_bounds__(64 * SWAVES) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_expand_a(const uint8_t* __restrict__ rho, size_t rho_stride,
                                                          const uint32_t* __restrict__ key_idx,
                                                          int32_t* __restrict__ a_hat, size_t n_ops, uint32_t n_keys) {
    __shared__ uint32_t lds[SWAVES * 64 * STAGE_STRIDE + 4];  // + keep_leftover's read-ahead past the last row
    __shared__ uint32_t meta_lds[SWAVES * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* stage = lds + wave * 64 * STAGE_STRIDE;
    uint32_t* meta = meta_lds + wave * 64;
    uint32_t* my = stage + lane * STAGE_STRIDE;
    const size_t g = (size_t)blockIdx.x * (64 * SWAVES) + threadIdx.x;
    const size_t wave_base = g - lane;
    const size_t n_streams = n_ops * (size_t)(K * L);
    const bool valid = g < n_streams;
    const size_t op = valid ? g / (K * L) : 0;
    const int rs = valid ? (int)(g % (K * L)) : 0;
    const int r = rs / L, sidx = rs % L;

    KeccakState st;
    // n_keys != 0: key_idx comes unchecked from the caller (verify_batch checks it beside this kernel, k_sanitize_keys on the
    // second stream): an out-of-range index reads key 0 here, its op is refused there
    size_t key = key_idx ? key_idx[op] : op;
    if (n_keys && key >= n_keys) key = 0;
    expand_a_seed(st, rho + key * rho_stride, sidx, r);
    if constexpr (PACK24) {
        (void)stage; (void)meta; (void)my;
        rej_ntt_poly_lane_direct(st, a_hat, wave_base, lane, valid);
    } else {
        rej_ntt_poly_lane(st, stage, meta, my, a_hat, wave_base, lane, valid);
    }
}

// ------------------------------------------------------------------------------------
// ExpandS (hashing.rs:252-272) = (L + K) x RejBoundedPoly (hashing.rs:158-213):
// stream (op, r): SHAKE256(rho' || r || 0), r < L -> s1[r], r >= L -> s2[r - L]; each byte
// gives two half-byte candidates (coeff_from_half_byte, conversion.rs:80-111).
// Output polys [op][L + K] (s1 then s2) with coefficients in [-eta, eta].
// S8 (key generation's own s1 / s2): one BYTE per coefficient, 256 bytes per polynomial in coefficient order -- the staging
// row holds bytes too (132 of them), so a 136-byte block is flushed in three groups of up to 96 candidates instead of twelve of 24,
// in 16-byte pieces; the seam-level mldsa_expand_s keeps int32[256].
template <int ETA, bool S8 = false>
__global__ __launch_bounds__(64 * SWAVES) void k_expand_s(const uint8_t* __restrict__ rho_prime, size_t rho_stride,
                                                          int32_t* __restrict__ s12, int polys_per_op, size_t n_ops) {
    __shared__ uint32_t lds[SWAVES * 64 * STAGE_STRIDE + 4];  // + keep_leftover's read-ahead past the last row
    __shared__ uint32_t meta_lds[SWAVES * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* stage = lds + wave * 64 * STAGE_STRIDE;
    uint32_t* meta = meta_lds + wave * 64;
    uint32_t* my = stage + lane * STAGE_STRIDE;
    const size_t g = (size_t)blockIdx.x * (64 * SWAVES) + threadIdx.x;
    const size_t wave_base = g - lane;
    const size_t n_streams = n_ops * (size_t)polys_per_op;
    const bool valid = g < n_streams;
    const size_t op = valid ? g / polys_per_op : 0;
    const uint32_t r = valid ? (uint32_t)(g % polys_per_op) : 0;

    KeccakState st;
    keccak_zero(st);
    {
        absorb_words<8>(st, rho_prime + op * rho_stride);
        st.lo[8] = r | (0x1Fu << 16);  // hashing.rs:260/266: rho' || r || 0  (then pad)
        st.hi[SHAKE256_RATE / 8 - 1] = 0x80000000u;
    }
    int n = valid ? 0 : N;  // coefficients already in s12 (a multiple of 4; S8: of 16)
    int carry = 0;          // accepted coefficients waiting in my[0 .. carry), < 4 (S8: < 16)
    if constexpr (S8) {
        uint8_t* myb = reinterpret_cast<uint8_t*>(my);
        uint8_t* out8 = reinterpret_cast<uint8_t*>(s12);
        while (__any(n < N)) {
            keccak_f1600(st);
            static_for<0, 3>([&](auto gc) {  // words 0..11, 12..23, 24..33 of the block: at most 15 + 96 bytes in the row
                constexpr int G = decltype(gc)::value;
                constexpr int NW = (G == 2) ? 10 : 12;
                int cnt = carry;
                static_for<0, NW>([&](auto wc) {
                    constexpr int W = 12 * G + decltype(wc)::value;
                    const uint32_t w = state_word<W>(st);
#pragma unroll
                    for (int k = 0; k < 8; k++) {  // low nibble of each byte first (hashing.rs:177-180)
                        int32_t v;
                        const bool ok = half_byte<ETA>((w >> (4 * k)) & 15u, v);
                        myb[cnt] = (uint8_t)v;
                        cnt += ok ? 1 : 0;
                    }
                });
                const int have = min(cnt, N - n);
                const int fc = (n + have == N) ? have : (have & ~15);  // bytes to flush: a multiple of 16 (N and n are)
                meta[lane] = ((uint32_t)fc << 16) | (uint32_t)n;
                wave_lds_sync();
                {
                    const int grp = lane >> 3, j16 = (lane & 7) * 16;
                    const size_t wb = ((size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_base >> 32)) << 32) |
                                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wave_base);
                    uint8_t* base = out8 + wb * N;
                    uint32_t m[8];
#pragma unroll
                    for (int i = 0; i < 8; i++) m[i] = meta[8 * i + grp];
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int row = 8 * i + grp;
                        if (j16 < (int)(m[i] >> 16)) {
                            const uint32_t* src = stage + row * STAGE_STRIDE + (lane & 7) * 4;
                            *reinterpret_cast<uint4*>(base + (uint32_t)row * N + (m[i] & 0xFFFFu) + j16) = make_uint4(src[0], src[1], src[2], src[3]);
                        }
                    }
                }
                wave_lds_sync();
                {  // the (at most 15) unflushed bytes to the row front
                    const uint32_t* q = my + (fc >> 2);
                    const uint32_t a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
                    my[0] = a0; my[1] = a1; my[2] = a2; my[3] = a3;
                }
                carry = have - fc;
                n += fc;
            });
        }
        return;
    }
    while (__any(n < N)) {
        keccak_f1600(st);
        // 136 bytes = 34 words; 3 words (24 half-bytes) per flush so that carry + candidates fit a row
        static_for<0, 12>([&](auto gc) {
            constexpr int G = decltype(gc)::value;
            constexpr int NW = (G == 11) ? 1 : 3;
            int cnt = carry;
            static_for<0, NW>([&](auto wc) {
                constexpr int W = 3 * G + decltype(wc)::value;
                const uint32_t w = state_word<W>(st);
#pragma unroll
                for (int k = 0; k < 8; k++) {  // low nibble of each byte first (hashing.rs:177-180)
                    int32_t v;
                    const bool ok = half_byte<ETA>((w >> (4 * k)) & 15u, v);
                    my[cnt] = (uint32_t)v;
                    cnt += ok ? 1 : 0;
                }
            });
            const int have = min(cnt, N - n);
            const int fc = (n + have == N) ? have : (have & ~3);
            flush_rows4(stage, meta, s12, wave_base, fc, n, lane);
            keep_leftover(my, fc);
            carry = have - fc;
            n += fc;
        });
    }
}

// ------------------------------------------------------------------------------------
// ExpandMask (hashing.rs:281-313): stream (op, r < L): v = SHAKE256(rho'' || (kappa + r) LE16),
// y[r][i] = gamma1 - (c-bit field i of v), c = 1 + bitlen(gamma1 - 1) (bit_unpack,
// conversion.rs:227-262).  No rejection: all lanes advance in lock step.
// RAW (the signer's rounds): y is kept as the squeezed bytes themselves -- 32 c bytes per polynomial (576 / 640 instead of 1 024),
// stored from the state registers, each lane to its own row (no LDS staging: ExpandMask 1.46 -> 1.41 ms per 65 536-op ML-DSA-65
// signing call); BitUnpack happens where y is used (sign_w's forward transforms, k_sign_tail / k_resolve), and the flags of the
// polynomials that can fail the ||z|| test (yrisk) come from sign_w.
template <int GB, bool RAW = false>  // gamma1 = 2^GB, GB = 17 or 19
__global__ __launch_bounds__(64 * SWAVES) __attribute__((amdgpu_waves_per_eu(4))) void k_expand_mask(const uint8_t* __restrict__ rho_pp, size_t rho_stride,
                                                             const uint16_t* __restrict__ kappa, int kappa_by_slot,
                                                             const uint32_t* __restrict__ op_idx,
                                                             int32_t* __restrict__ y, int l, size_t n_ops,
                                                             const uint32_t* __restrict__ n_dev,
                                                             uint8_t* __restrict__ yrisk, int32_t risk_bound) {
    constexpr int CB = GB + 1;
    constexpr uint32_t MASK = (1u << CB) - 1u;
    __shared__ uint32_t lds[SWAVES * 64 * STAGE_STRIDE + 4];  // + keep_leftover's read-ahead past the last row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* stage = lds + wave * 64 * STAGE_STRIDE;
    uint32_t* my = stage + lane * STAGE_STRIDE;
    if (n_dev) n_ops = *n_dev;  // the signer's rounds: slots of this round, known only on the device
    const size_t n_streams = n_ops * (size_t)l;
    // tiles of 256 streams, grid-stride (the grid is sized from the expected count)
    for (size_t tile = (size_t)blockIdx.x * (64 * SWAVES); tile < n_streams; tile += (size_t)gridDim.x * (64 * SWAVES)) {
    const size_t g = tile + threadIdx.x;
    const size_t wave_base = g - lane;
    const size_t wave_base_u = ((size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_base >> 32)) << 32) |
                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wave_base);  // the same value, in SGPRs
    const bool valid = g < n_streams;
    const size_t slot = valid ? g / l : 0;          // position in the (compacted) batch
    const uint32_t r = valid ? (uint32_t)(g % l) : 0;
    const size_t op = op_idx ? op_idx[slot] : slot; // which op's rho'' / kappa

    KeccakState st;
    keccak_zero(st);
    {
        absorb_words<8>(st, rho_pp + op * rho_stride);
        const uint32_t nn = ((uint32_t)kappa[kappa_by_slot ? slot : op] + r) & 0xFFFFu;  // hashing.rs:293 (u16 arithmetic)
        st.lo[8] = nn | (0x1Fu << 16);
        st.hi[SHAKE256_RATE / 8 - 1] = 0x80000000u;
    }
    // bit buffer; everything below is lane-uniform and resolves at compile time
    uint64_t acc = 0;
    int nbits = 0, n = 0, carry = 0;  // n = coefficients already stored (multiple of 4), carry < 4 wait in the row
    int32_t ymax = 0;
    // units of a stream's output row: coefficients, or (RAW) the dwords of its 32 c bytes
    constexpr int UNITS = RAW ? 8 * CB : N;
    if constexpr (RAW) {
        // the squeezed words straight from the registers to the stream's row: block b holds dwords 34 b .. 34 b + 33 of it, stored
        // as eight 16-byte pieces and one 8-byte piece (34 b is = 0 or 2 mod 4), each lane to its own row -- no staging, no flush
        uint32_t* row = reinterpret_cast<uint32_t*>(y) + g * (size_t)UNITS;
        static_for<0, 5>([&](auto bc) {
            constexpr int B = decltype(bc)::value;
            keccak_f1600(st);
            constexpr int W0 = (B & 1) ? 2 : 0;  // first word of the block whose row position is a multiple of four dwords
            if (valid) {
                if constexpr (W0 == 2 && 34 * B + 2 <= UNITS)
                    *reinterpret_cast<uint2*>(row + 34 * B) = make_uint2(state_word<0>(st), state_word<1>(st));
                static_for<0, 8>([&](auto qc) {
                    constexpr int W = W0 + 4 * decltype(qc)::value;
                    if constexpr (34 * B + W + 4 <= UNITS)
                        *reinterpret_cast<uint4*>(row + 34 * B + W) =
                            make_uint4(state_word<W>(st), state_word<W + 1>(st), state_word<W + 2>(st), state_word<W + 3>(st));
                });
                if constexpr (W0 == 0 && 34 * B + 34 <= UNITS)
                    *reinterpret_cast<uint2*>(row + 34 * B + 32) = make_uint2(state_word<32>(st), state_word<33>(st));
            }
        });
        continue;
    }
#pragma unroll
    for (int blk = 0; blk < 5; blk++) {
        keccak_f1600(st);
        static_for<0, 2>([&](auto hc) {
            constexpr int H = decltype(hc)::value;
            int cnt = carry;
            static_for<0, 17>([&](auto wc) {
                constexpr int W = 17 * H + decltype(wc)::value;
                if (n + cnt < N) {  // the reference squeezes 640 bytes but unpacks only 32*c (hashing.rs:297-301)
                    acc |= (uint64_t)state_word<W>(st) << nbits;
                    nbits += 32;
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        if (nbits >= CB && n + cnt < N) {
                            const int32_t yv = (1 << GB) - (int32_t)((uint32_t)acc & MASK);
                            my[cnt] = (uint32_t)yv;
                            ymax = max(ymax, yv < 0 ? -yv : yv);
                            cnt++;
                            acc >>= CB;
                            nbits -= CB;
                        }
                    }
                }
            });
            // uniform counts: every row holds cnt coefficients continuing at n; store the multiple-of-4
            // part as dwordx4 (8 lanes per row, 8 rows per store), keep the rest for the next flush
            const int fc = (n + cnt == UNITS) ? cnt : (cnt & ~3);
            if (fc > 0) {
                wave_lds_sync();
                const int grp = lane >> 3, j4 = (lane & 7) * 4;
                // wave-uniform base + 32-bit byte offsets (see flush_rows4)
                char* base = reinterpret_cast<char*>(y) + wave_base_u * (size_t)(UNITS * 4);
                const uint32_t lane_b = (uint32_t)(grp * UNITS + n + j4) * 4u;
                // rolled on purpose: unrolled, the eight row addresses stay live across the permutations and cost the
                // kernel its fourth wave per SIMD (142 -> 118 VGPRs)
#pragma unroll 1
                for (int i = 0; i < 8; i++) {
                    const int row = 8 * i + grp;
                    if (j4 < fc && wave_base + row < n_streams) {
                        const uint32_t* src = stage + row * STAGE_STRIDE + j4;
                        *reinterpret_cast<int4*>(base + (lane_b + (uint32_t)i * (8u * UNITS * 4u))) =
                            make_int4((int)src[0], (int)src[1], (int)src[2], (int)src[3]);
                    }
                }
                wave_lds_sync();
                if (cnt > fc) keep_leftover(my, fc);
            }
            carry = cnt - fc;
            n += fc;
        });
    }
    if constexpr (!RAW) {
        if (yrisk && valid) yrisk[g] = ymax >= risk_bound ? 1 : 0;
    }
    }
}

// ------------------------------------------------------------------------------------
// SampleInBall (hashing.rs:43-100): one op per lane.  SHAKE256(c_tilde): first 8 bytes =
// sign bits h; for i = 256 - tau .. 255: draw bytes j until j <= i; c[i] = c[j];
// c[j] = 1 - 2 * bit(i + tau - 256) of h.  The Fisher-Yates array and the squeezed block
// live in lane-private LDS rows (dynamic indexing); output c[op] as int32[256].
// C8 (the op-level pipelines): c as 256 bytes per op, the four coefficients 64 k + lane (each -1, 0 or 1) in lane's dword -- what a
// wave's forward transform loads; the seam-level mldsa_sample_in_ball keeps int32[256].
template <int CT, bool C8 = false>  // c_tilde bytes: 32, 48 or 64
__global__ __launch_bounds__(64) void k_sample_in_ball(const uint8_t* __restrict__ c_tilde, size_t ct_stride,
                                                       int tau, int32_t* __restrict__ c_out, size_t n_ops,
                                                       const uint32_t* __restrict__ n_dev) {
    __shared__ uint32_t c_lds[64 * SIB_C_STRIDE];
    __shared__ uint32_t b_lds[64 * SIB_BLK_STRIDE];
    const int lane = threadIdx.x;
    if (n_dev) n_ops = *n_dev;
    if constexpr (EXP_PRIO != 0) __builtin_amdgcn_s_setprio(EXP_PRIO);
    uint32_t* bw = b_lds + lane * SIB_BLK_STRIDE;
    for (size_t wave_base = (size_t)blockIdx.x * 64; wave_base < n_ops; wave_base += (size_t)gridDim.x * 64) {
        const size_t op = wave_base + lane;
        const bool valid = op < n_ops;

        KeccakState st;
        keccak_zero(st);
        if (valid) {
            absorb_words<CT / 8>(st, c_tilde + op * ct_stride);
        }
        shake_pad<SHAKE256_RATE, CT>(st);
        sample_in_ball_lane(st, tau, valid, c_lds + lane * SIB_C_STRIDE, bw);
        wave_lds_sync();
        for (int row = 0; row < 64; row++) {
            if (wave_base + row >= n_ops) break;
            if constexpr (C8) {
                const uint8_t* cb = reinterpret_cast<const uint8_t*>(c_lds + row * SIB_C_STRIDE);
                reinterpret_cast<uint32_t*>(c_out)[(wave_base + row) * 64 + lane] =
                    (uint32_t)cb[lane] | ((uint32_t)cb[64 + lane] << 8) | ((uint32_t)cb[128 + lane] << 16) | ((uint32_t)cb[192 + lane] << 24);
                continue;
            }
            const uint32_t packed = c_lds[row * SIB_C_STRIDE + lane];
            int4 v = make_int4((int8_t)(packed & 0xFF), (int8_t)((packed >> 8) & 0xFF), (int8_t)((packed >> 16) & 0xFF),
                               (int8_t)(packed >> 24));
            reinterpret_cast<int4*>(c_out + (wave_base + row) * N)[lane] = v;
        }
        wave_lds_sync();
    }
}

// ------------------------------------------------------------------------- launchers
static inline unsigned stream_blocks(size_t n_streams) { return (unsigned)((n_streams + 64 * SWAVES - 1) / (64 * SWAVES)); }

// ExpandA (24-bit form) for SMALL calls: one polynomial per WAVE on the interleaved cooperative sponge (keccak_coop2.h: 2.2 us per
// permutation; expand_coop_dev.h expand_a_coop2_poly).  After every permutation the lanes holding the 21 rate words put the 168-byte block
// into the wave's LDS row; lanes 0-55 test its 56 candidates (coeff_from_three_bytes, conversion.rs:40-61), rank the accepted ones with a
// ballot and store each as its three bytes at 3 i of the polynomial's 768-byte row -- the same bytes k_expand_a<.., true> writes.
template <int K, int L>
__global__ __launch_bounds__(64 * SWAVES) void k_expand_a_coop(const uint8_t* __restrict__ rho, size_t rho_stride, const uint32_t* __restrict__ key_idx,
                                                               int32_t* __restrict__ a_hat, size_t n_ops, uint32_t n_keys) {
    __shared__ uint32_t blk_lds[SWAVES * EA_COOP_BLK_DWORDS];
    const int lane = threadIdx.x & 63;
    const Coop2Lane c = coop2_lane(lane);
    uint32_t* blk = blk_lds + (threadIdx.x >> 6) * EA_COOP_BLK_DWORDS;
    const size_t n_streams = n_ops * (size_t)(K * L);
    const size_t wave0 = (size_t)blockIdx.x * SWAVES + (threadIdx.x >> 6), stride = (size_t)gridDim.x * SWAVES;
    for (size_t g = wave0; g < n_streams; g += stride)  // wave-uniform
        expand_a_coop2_poly<K, L>(rho, rho_stride, key_idx, a_hat, g, n_keys, blk, lane, c);
}

int launch_expand_a(mldsa_ctx* ctx, int set, const uint8_t* rho, size_t rho_stride, const uint32_t* key_idx, int32_t* a_hat,
                    size_t n_ops, hipStream_t s, bool pack24, size_t n_keys_unchecked) {
    const uint32_t n_keys = (uint32_t)std::min<size_t>(n_keys_unchecked, 0xFFFFFFFFu);
    if (n_ops == 0) return MLDSA_OK;
    const mldsa_params* p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "expand_a: unknown parameter set");
    dim3 grid(stream_blocks(n_ops * (size_t)(p->k * p->l))), block(64 * SWAVES);
    if (pack24 && ctx->opt_coop_hash && n_ops * (size_t)(p->k * p->l) <= ctx->coop_a_max) {  // a small call: all latency
        const dim3 cgrid((unsigned)((n_ops * (size_t)(p->k * p->l) + SWAVES - 1) / SWAVES));
        if (set == MLDSA_44) hipLaunchKernelGGL((k_expand_a_coop<4, 4>), cgrid, block, 0, s, rho, rho_stride, key_idx, a_hat, n_ops, n_keys);
        else if (set == MLDSA_65) hipLaunchKernelGGL((k_expand_a_coop<6, 5>), cgrid, block, 0, s, rho, rho_stride, key_idx, a_hat, n_ops, n_keys);
        else hipLaunchKernelGGL((k_expand_a_coop<8, 7>), cgrid, block, 0, s, rho, rho_stride, key_idx, a_hat, n_ops, n_keys);
        MLDSA_HIP_CHECK(hipGetLastError());
        return MLDSA_OK;
    }
#define MLDSA_EA(KK, LL)                                                                                                          \
    do {                                                                                                                          \
        if (pack24) hipLaunchKernelGGL((k_expand_a<KK, LL, true>), grid, block, 0, s, rho, rho_stride, key_idx, a_hat, n_ops, n_keys);     \
        else hipLaunchKernelGGL((k_expand_a<KK, LL, false>), grid, block, 0, s, rho, rho_stride, key_idx, a_hat, n_ops, n_keys);          \
    } while (0)
    if (set == MLDSA_44) MLDSA_EA(4, 4);
    else if (set == MLDSA_65) MLDSA_EA(6, 5);
    else MLDSA_EA(8, 7);
#undef MLDSA_EA
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// ExpandS (one byte per coefficient: key generation's own s1 / s2) for SMALL calls: one polynomial per wave (keccak_coop2.h).  The
// 136-byte block goes through the wave's LDS row; its 272 half-byte candidates (low nibble first, hashing.rs:177-180) are tested in
// five passes of 64, ranked with a ballot and stored at their coefficient index -- the bytes k_expand_s<ETA, true> writes.
template <int ETA>
__global__ __launch_bounds__(64 * SWAVES) void k_expand_s_coop(const uint8_t* __restrict__ rho_prime, size_t rho_stride, int32_t* __restrict__ s12,
                                                               int polys_per_op, size_t n_ops) {
    __shared__ uint32_t blk_lds[SWAVES * ES_COOP_BLK_DWORDS];
    const int lane = threadIdx.x & 63;
    const Coop2Lane c = coop2_lane(lane);
    uint32_t* blk = blk_lds + (threadIdx.x >> 6) * ES_COOP_BLK_DWORDS;
    const size_t n_streams = n_ops * (size_t)polys_per_op;
    const size_t wave0 = (size_t)blockIdx.x * SWAVES + (threadIdx.x >> 6), stride = (size_t)gridDim.x * SWAVES;
    for (size_t g = wave0; g < n_streams; g += stride) {  // wave-uniform
        const size_t op = g / polys_per_op;
        expand_s_coop2_poly<ETA>(rho_prime + op * rho_stride, (uint32_t)(g % polys_per_op), reinterpret_cast<uint8_t*>(s12) + g * (size_t)N, blk, lane, c);
    }
}

int launch_expand_s(mldsa_ctx* ctx, int set, const uint8_t* rho_prime, size_t rho_stride, int32_t* s12, size_t n_ops, hipStream_t s, bool s8) {
    if (n_ops == 0) return MLDSA_OK;
    const mldsa_params* p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "expand_s: unknown parameter set");
    const int ppo = p->k + p->l;
    dim3 grid(stream_blocks(n_ops * (size_t)ppo)), block(64 * SWAVES);
    if (s8 && ctx->opt_coop_hash && n_ops * (size_t)ppo <= ctx->coop_a_max) {  // a small key generation: all latency
        const dim3 cgrid((unsigned)((n_ops * (size_t)ppo + SWAVES - 1) / SWAVES));
        if (p->eta == 2) hipLaunchKernelGGL((k_expand_s_coop<2>), cgrid, block, 0, s, rho_prime, rho_stride, s12, ppo, n_ops);
        else hipLaunchKernelGGL((k_expand_s_coop<4>), cgrid, block, 0, s, rho_prime, rho_stride, s12, ppo, n_ops);
        MLDSA_HIP_CHECK(hipGetLastError());
        return MLDSA_OK;
    }
    if (s8) {
        if (p->eta == 2) hipLaunchKernelGGL((k_expand_s<2, true>), grid, block, 0, s, rho_prime, rho_stride, s12, ppo, n_ops);
        else hipLaunchKernelGGL((k_expand_s<4, true>), grid, block, 0, s, rho_prime, rho_stride, s12, ppo, n_ops);
        MLDSA_HIP_CHECK(hipGetLastError());
        return MLDSA_OK;
    }
    if (p->eta == 2) hipLaunchKernelGGL((k_expand_s<2>), grid, block, 0, s, rho_prime, rho_stride, s12, ppo, n_ops);
    else hipLaunchKernelGGL((k_expand_s<4>), grid, block, 0, s, rho_prime, rho_stride, s12, ppo, n_ops);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// RAW ExpandMask for SMALL rounds: one stream per wave (keccak_coop2.h: 2.2 instead of 9.4 us per permutation of a latency-bound
// launch).  Same arguments and the same bytes as k_expand_mask<GB, true>: the lanes holding state words 0 .. 16 store their half
// (the E lane the word's low dword, the O lane the high one) of every squeezed block straight into the stream's row of 32 c bytes.
template <int GB>
__global__ __launch_bounds__(64 * SWAVES) void k_expand_mask_coop(const uint8_t* __restrict__ rho_pp, size_t rho_stride, const uint16_t* __restrict__ kappa,
                                                                  int kappa_by_slot, const uint32_t* __restrict__ op_idx, int32_t* __restrict__ y, int l,
                                                                  size_t n_ops, const uint32_t* __restrict__ n_dev) {
    constexpr int ROW_BYTES = 32 * (GB + 1);
    const int lane = threadIdx.x & 63;
    const Coop2Lane c = coop2_lane(lane);
    if (n_dev) n_ops = *n_dev;
    const size_t n_streams = n_ops * (size_t)l;
    const size_t wave0 = (size_t)blockIdx.x * SWAVES + (threadIdx.x >> 6), stride = (size_t)gridDim.x * SWAVES;
    for (size_t g = wave0; g < n_streams; g += stride) {  // wave-uniform
        const size_t slot = g / l;
        const uint32_t r = (uint32_t)(g % l);
        const size_t op = op_idx ? op_idx[slot] : slot;
        expand_mask_coop2_poly<GB>(rho_pp + op * rho_stride, (uint32_t)kappa[kappa_by_slot ? slot : op] + r, reinterpret_cast<uint8_t*>(y) + g * (size_t)ROW_BYTES, lane, c);
    }
}

// n_dev != nullptr: the op count is read from the device (the signer's rounds) and n_ops only sizes the grid
int launch_expand_mask(mldsa_ctx* ctx, int set, const uint8_t* rho_pp, size_t rho_stride, const uint16_t* kappa, int kappa_by_slot,
                       const uint32_t* op_idx, int32_t* y, size_t n_ops, hipStream_t s, uint8_t* yrisk, const uint32_t* n_dev, bool raw) {
    if (n_ops == 0 && !n_dev) return MLDSA_OK;
    const mldsa_params* p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "expand_mask: unknown parameter set");
    dim3 grid(stream_blocks((n_ops ? n_ops : 1) * (size_t)p->l)), block(64 * SWAVES);
    if (raw && ctx->opt_coop_hash && (n_ops ? n_ops : 1) * (size_t)p->l <= ctx->coop_mask_max) {  // a small round: all latency
        const dim3 cgrid((unsigned)(((n_ops ? n_ops : 1) * (size_t)p->l + SWAVES - 1) / SWAVES));
        if (p->gamma1 == (1 << 17)) hipLaunchKernelGGL((k_expand_mask_coop<17>), cgrid, block, 0, s, rho_pp, rho_stride, kappa, kappa_by_slot, op_idx, y, p->l, n_ops, n_dev);
        else hipLaunchKernelGGL((k_expand_mask_coop<19>), cgrid, block, 0, s, rho_pp, rho_stride, kappa, kappa_by_slot, op_idx, y, p->l, n_ops, n_dev);
        MLDSA_HIP_CHECK(hipGetLastError());
        return MLDSA_OK;
    }
    if (raw) {
        if (p->gamma1 == (1 << 17)) hipLaunchKernelGGL((k_expand_mask<17, true>), grid, block, 0, s, rho_pp, rho_stride, kappa, kappa_by_slot, op_idx, y, p->l, n_ops, n_dev, yrisk, 0);
        else hipLaunchKernelGGL((k_expand_mask<19, true>), grid, block, 0, s, rho_pp, rho_stride, kappa, kappa_by_slot, op_idx, y, p->l, n_ops, n_dev, yrisk, 0);
        MLDSA_HIP_CHECK(hipGetLastError());
        return MLDSA_OK;
    }
    if (p->gamma1 == (1 << 17)) hipLaunchKernelGGL((k_expand_mask<17>), grid, block, 0, s, rho_pp, rho_stride, kappa, kappa_by_slot, op_idx, y, p->l, n_ops, n_dev, yrisk, p->gamma1 - 2 * p->beta);
    else hipLaunchKernelGGL((k_expand_mask<19>), grid, block, 0, s, rho_pp, rho_stride, kappa, kappa_by_slot, op_idx, y, p->l, n_ops, n_dev, yrisk, p->gamma1 - 2 * p->beta);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// SampleInBall (one byte per coefficient) for SMALL calls: one op per wave (challenge_dev.h sample_in_ball_coop2: cooperative sponge, the
// Fisher-Yates walk of hashing.rs:68-83 wave-uniform in registers).  The bytes k_sample_in_ball<.., true> writes.
template <int CT>
__global__ __launch_bounds__(64 * SWAVES) void k_sample_in_ball_coop(const uint8_t* __restrict__ c_tilde, size_t ct_stride, int tau, int32_t* __restrict__ c_out,
                                                                     size_t n_ops, const uint32_t* __restrict__ n_dev) {
    __shared__ uint32_t rows[SWAVES * 36];
    const int lane = threadIdx.x & 63;
    const Coop2Lane c = coop2_lane(lane);
    uint32_t* bw = rows + (threadIdx.x >> 6) * 36;
    if (n_dev) n_ops = *n_dev;
    const size_t wave0 = (size_t)blockIdx.x * SWAVES + (threadIdx.x >> 6), stride = (size_t)gridDim.x * SWAVES;
    for (size_t op = wave0; op < n_ops; op += stride)  // wave-uniform
        reinterpret_cast<uint32_t*>(c_out)[op * 64 + lane] = sample_in_ball_coop2<CT>(c_tilde + op * ct_stride, tau, bw, lane, c);
}

int launch_sample_in_ball(mldsa_ctx* ctx, int set, const uint8_t* c_tilde, size_t ct_stride, int32_t* c, size_t n_ops, hipStream_t s,
                          const uint32_t* n_dev, bool c8) {
    if (n_ops == 0 && !n_dev) return MLDSA_OK;
    const mldsa_params* p = params_of(set);
    if (!p) return set_error(MLDSA_ERR_PARAM, "sample_in_ball: unknown parameter set");
    dim3 grid((unsigned)(((n_ops ? n_ops : 1) + 63) / 64)), block(64);
    if (c8 && ctx && ctx->opt_coop_hash && (n_ops ? n_ops : 1) <= ctx->coop_sib_max) {  // a small call / round: one op per wave
        const dim3 cgrid((unsigned)(((n_ops ? n_ops : 1) + SWAVES - 1) / SWAVES)), cblock(64 * SWAVES);
        if (p->ctilde_len == 32) hipLaunchKernelGGL((k_sample_in_ball_coop<32>), cgrid, cblock, 0, s, c_tilde, ct_stride, p->tau, c, n_ops, n_dev);
        else if (p->ctilde_len == 48) hipLaunchKernelGGL((k_sample_in_ball_coop<48>), cgrid, cblock, 0, s, c_tilde, ct_stride, p->tau, c, n_ops, n_dev);
        else hipLaunchKernelGGL((k_sample_in_ball_coop<64>), cgrid, cblock, 0, s, c_tilde, ct_stride, p->tau, c, n_ops, n_dev);
        MLDSA_HIP_CHECK(hipGetLastError());
        return MLDSA_OK;
    }
    if (c8) {
        if (p->ctilde_len == 32) hipLaunchKernelGGL((k_sample_in_ball<32, true>), grid, block, 0, s, c_tilde, ct_stride, p->tau, c, n_ops, n_dev);
        else if (p->ctilde_len == 48) hipLaunchKernelGGL((k_sample_in_ball<48, true>), grid, block, 0, s, c_tilde, ct_stride, p->tau, c, n_ops, n_dev);
        else hipLaunchKernelGGL((k_sample_in_ball<64, true>), grid, block, 0, s, c_tilde, ct_stride, p->tau, c, n_ops, n_dev);
        MLDSA_HIP_CHECK(hipGetLastError());
        return MLDSA_OK;
    }
    if (p->ctilde_len == 32) hipLaunchKernelGGL((k_sample_in_ball<32>), grid, block, 0, s, c_tilde, ct_stride, p->tau, c, n_ops, n_dev);
    else if (p->ctilde_len == 48) hipLaunchKernelGGL((k_sample_in_ball<48>), grid, block, 0, s, c_tilde, ct_stride, p->tau, c, n_ops, n_dev);
    else hipLaunchKernelGGL((k_sample_in_ball<64>), grid, block, 0, s, c_tilde, ct_stride, p->tau, c, n_ops, n_dev);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

}  // namespace mldsa
