// Device-side building blocks of the sampler kernels (kernels_sample.hip; static_for and the block accessors are also used
// by the hash kernels of kernels_codec.hip): compile-time loops, static access to the squeezed block,
// lane-private LDS staging rows and their cooperative, coalesced flush.
#pragma once
#include <type_traits>

#include "field.h"
#include "keccak.h"

namespace mldsa {

constexpr int SWAVES = 4;         // waves per block for the sampler kernels
constexpr int STAGE_STRIDE = 33;  // dwords per lane row (odd: conflict-free), capacity 32

template <int I, int E, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

// 32-bit word W (compile time) of the sponge state
template <int W>
__device__ __forceinline__ uint32_t state_word(const KeccakState& s) {
    if constexpr (W & 1) return s.hi[W / 2]; else return s.lo[W / 2];
}

// 32 bits of the squeezed block starting at byte B (compile time)
template <int B>
__device__ __forceinline__ uint32_t block_bits(const KeccakState& s) {
    constexpr int W = B / 4, SH = (B % 4) * 8;
    if constexpr (SH == 0) return state_word<W>(s);
    else if constexpr (W + 1 < 50) return __builtin_amdgcn_alignbit(state_word<W + 1>(s), state_word<W>(s), SH);
    else return state_word<W>(s) >> SH;
}

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Flush in 16-byte pieces: lane row `r` holds fc[r] dwords (a multiple of 4) that continue stream
// r's polynomial at coefficient n[r] (a multiple of 4).  Eight lanes serve one row, so one
// iteration stores 8 rows x 32 coefficients with one dwordx4 store per lane: 8 iterations per
// flush instead of 32.  Rows keep up to 3 left-over coefficients for the next flush (the caller
// moves them to the front of its row).
// (int32 rows: the seam-level mldsa_expand_a / mldsa_expand_s; the pipelines' own 24-bit A_hat is written without staging,
// rej_ntt_poly_lane_direct below.)
__device__ __forceinline__ void flush_rows4(uint32_t* stage, uint32_t* meta, int32_t* __restrict__ out, size_t wave_base,
                                            int fc, int n, int lane) {
    constexpr uint32_t POLY_BYTES = N * 4;
    // per row: fc << 16 | byte offset of coefficient n inside the stream's polynomial
    meta[lane] = ((uint32_t)fc << 16) | (uint32_t)(n * 4);
    wave_lds_sync();
    const int grp = lane >> 3, j4 = (lane & 7) * 4;
    // The wave's 64 polynomials are contiguous and start at a wave-uniform address: scalar base + 32-bit byte offset per
    // store (global_store ... v_off, s[base]) instead of a 64-bit address per row kept in registers or rebuilt each time.
    const size_t wb = ((size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_base >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wave_base);
    char* base = reinterpret_cast<char*>(out) + wb * POLY_BYTES;
    const uint32_t lane_b = (uint32_t)grp * POLY_BYTES + (uint32_t)(lane & 7) * 16u;
    uint32_t m[8];
#pragma unroll
    for (int i = 0; i < 8; i++) m[i] = meta[8 * i + grp];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int row = 8 * i + grp;
        if (j4 < (int)(m[i] >> 16)) {
            const uint32_t* src = stage + row * STAGE_STRIDE + j4;
            const uint32_t off = lane_b + (uint32_t)i * (8u * POLY_BYTES) + (m[i] & 0xFFFFu);
            *reinterpret_cast<int4*>(base + off) = make_int4((int)src[0], (int)src[1], (int)src[2], (int)src[3]);
        }
    }
    wave_lds_sync();
}

// after flush_rows4: move the (at most 3) unflushed coefficients my[fc .. fc + 2] to the row front
__device__ __forceinline__ void keep_leftover(uint32_t* my, int fc) {
    const uint32_t a = my[fc], b = my[fc + 1], c = my[fc + 2];
    my[0] = a;
    my[1] = b;
    my[2] = c;
}

// RejNTTPoly (hashing.rs:111-146) for the lane's stream: `st` holds the absorbed, padded seed.
// Squeezes SHAKE128 blocks until the lane has its 256 coefficients (wave-uniform loop: every lane
// keeps permuting until the whole wave is done, extra output is dropped) and flushes 28-candidate
// half blocks through the staging rows to out[(wave_base + lane) * 256 + ...].
__device__ __forceinline__ void rej_ntt_poly_lane(KeccakState& st, uint32_t* stage, uint32_t* meta, uint32_t* my,
                                                  int32_t* __restrict__ out, size_t wave_base, int lane, bool valid) {
    int n = valid ? 0 : N;  // coefficients already in `out` (a multiple of 4)
    int carry = 0;          // accepted coefficients waiting in my[0 .. carry), < 4
    while (__any(n < N)) {
        keccak_f1600(st);
        static_for<0, 2>([&](auto hc) {
            constexpr int H = decltype(hc)::value;
            int cnt = carry;
            static_for<0, 28>([&](auto cc) {  // 28 candidates per half block: bytes 84 H + 3 C
                constexpr int C = decltype(cc)::value;
                const uint32_t z = block_bits<84 * H + 3 * C>(st) & 0x7FFFFFu;  // coeff_from_three_bytes, conversion.rs:40-61
                my[cnt] = z;
                cnt += (z < (uint32_t)Q) ? 1 : 0;
            });
            const int have = min(cnt, N - n);
            const int fc = (n + have == N) ? have : (have & ~3);  // N and n are multiples of 4, so fc is too
            flush_rows4(stage, meta, out, wave_base, fc, n, lane);
            keep_leftover(my, fc);
            carry = have - fc;
            n += fc;
        });
    }
}

// The PACK24 form of the same sampler, WITHOUT staging: the packed polynomial is the squeezed byte stream itself with bit 23 of every
// 3-byte candidate cleared and the rejected candidates (1 in 1 024) left out.  A lane walks its block in groups of four candidates =
// three state words: if all four are below q (and four are still wanted) the three masked words go to its row with one 12-byte
// store at byte 3 x count -- the lane's own row, any alignment -- else the accepted ones are stored byte by byte.  A rejection in
// some lane of the wave happens in about a fifth of the groups; only those groups run the byte path.  No LDS, no flush, and a row's
// lines fill from consecutive stores a few cycles apart instead of from two flushes ~10 us apart.
struct __attribute__((packed, aligned(1))) Packed3Unaligned { uint32_t a, b, c; };

__device__ __forceinline__ void rej_ntt_poly_lane_direct(KeccakState& st, int32_t* __restrict__ out, size_t wave_base, int lane, bool valid) {
    constexpr uint32_t ROW = PACKED_POLY_DWORDS * 4;
    // wave-uniform base + a 32-bit byte offset per lane (global_store ... v_off, s[base]): `off` = lane's row + bytes stored so far
    const size_t wb = ((size_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(wave_base >> 32)) << 32) |
                      (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wave_base);
    uint8_t* base = reinterpret_cast<uint8_t*>(out) + wb * ROW;
    const uint32_t end = (uint32_t)(lane + 1) * ROW;   // the row is full when off reaches this
    uint32_t off = valid ? (uint32_t)lane * ROW : end;
    while (__any(off < end)) {
        keccak_f1600(st);
        static_for<0, 14>([&](auto gc) {  // group G: bytes 12 G .. 12 G + 11 of the 168-byte block = four candidates
            constexpr int G = decltype(gc)::value;
            // coeff_from_three_bytes (conversion.rs:40-61) x 4, in place: bit 23 of every 3-byte field cleared ...
            const uint32_t m0 = state_word<3 * G>(st) & 0xFF7FFFFFu, m1 = state_word<3 * G + 1>(st) & 0xFFFF7FFFu,
                           m2 = state_word<3 * G + 2>(st) & 0x7FFFFF7Fu;
            // ... and z >= q  <=>  z + (2^23 - q) carries into the cleared bit: one 96-bit addition of 0x001FFF per field (no field
            // carries into its neighbour: z + 8191 < 2^24) and a test of the four bits 23
            const uint64_t lo = (((uint64_t)m1 << 32) | m0) + 0x1FFF001FFF001FFFull;
            const uint32_t hi = m2 + 0x001FFF00u + (uint32_t)(lo < 0x1FFF001FFF001FFFull ? 1u : 0u);
            const uint32_t over = ((uint32_t)lo & 0x00800000u) | ((uint32_t)(lo >> 32) & 0x00008000u) | (hi & 0x80000080u);
            if (over == 0u && off + 12u <= end) {
                *reinterpret_cast<Packed3Unaligned*>(base + off) = Packed3Unaligned{m0, m1, m2};
                off += 12u;
            } else if (off < end) {
                const uint32_t c[4] = {m0 & 0x7FFFFFu, __builtin_amdgcn_alignbit(m1, m0, 24) & 0x7FFFFFu,
                                       __builtin_amdgcn_alignbit(m2, m1, 16) & 0x7FFFFFu, m2 >> 8};
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    if (c[k] < (uint32_t)Q && off < end) {
                        uint8_t* dst = base + off;
                        dst[0] = (uint8_t)c[k];
                        dst[1] = (uint8_t)(c[k] >> 8);
                        dst[2] = (uint8_t)(c[k] >> 16);
                        off += 3u;
                    }
                }
            }
        });
    }
}

// absorbed + padded SHAKE128 state of stream (rho, s, r)   (hashing.rs:236: rho || s || r)
__device__ __forceinline__ void expand_a_seed(KeccakState& st, const uint8_t* rho, int s_idx, int r_idx) {
    keccak_zero(st);
    absorb_words<4>(st, rho);
    st.lo[4] = (uint32_t)s_idx | ((uint32_t)r_idx << 8) | (0x1Fu << 16);
    st.hi[SHAKE128_RATE / 8 - 1] = 0x80000000u;
}

}  // namespace mldsa
