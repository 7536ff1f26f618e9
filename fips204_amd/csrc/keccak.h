// Keccak-f[1600] / SHAKE128 / SHAKE256 for gfx950, one sponge state per lane
// (64 independent XOF streams per wavefront, state in VGPRs as 25 x (lo, hi) 32-bit pairs).
//
// Replaces the third-party `sha3` crate the reference calls through h256_xof / g128_xof
// (src/hashing.rs:13-27; FIPS 202).  Lane-per-state was chosen over a wave-cooperative
// (25/50 lanes per state) layout because theta/pi would then be cross-lane traffic: about
// 4x more issue slots per permutation (DESIGN.md "Keccak layout").
//
// 64-bit rotates are two v_alignbit_b32; chi and the 5-way theta parities are v_bitop3_b32
// (measured: 190 VALU per round, 70 VGPRs, no scratch).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mldsa {

constexpr int SHAKE128_RATE = 168;
constexpr int SHAKE256_RATE = 136;

struct KeccakState {
    uint32_t lo[25];
    uint32_t hi[25];
};

__device__ __constant__ const uint32_t KECCAK_RC_LO[24] = {
    0x00000001u, 0x00008082u, 0x0000808au, 0x80008000u, 0x0000808bu, 0x80000001u, 0x80008081u, 0x00008009u,
    0x0000008au, 0x00000088u, 0x80008009u, 0x8000000au, 0x8000808bu, 0x0000008bu, 0x00008089u, 0x00008003u,
    0x00008002u, 0x00000080u, 0x0000800au, 0x8000000au, 0x80008081u, 0x00008080u, 0x80000001u, 0x80008008u};
// high words of the round constants are either 0 or 0x80000000: one bit per round
constexpr uint32_t KECCAK_RC_HI_BITS = 0x00BBE0CCu;  // bit r set <=> RC[r] >> 63

__device__ __forceinline__ void keccak_zero(KeccakState& s) {
#pragma unroll
    for (int i = 0; i < 25; i++) { s.lo[i] = 0; s.hi[i] = 0; }
}

// rotl64 of (hi:lo) by a compile-time amount
template <int R>
__device__ __forceinline__ void rotl64(uint32_t lo, uint32_t hi, uint32_t& olo, uint32_t& ohi) {
    if constexpr (R == 0) {
        olo = lo; ohi = hi;
    } else if constexpr (R == 32) {
        olo = hi; ohi = lo;
    } else if constexpr (R < 32) {
        ohi = __builtin_amdgcn_alignbit(hi, lo, 32 - R);
        olo = __builtin_amdgcn_alignbit(lo, hi, 32 - R);
    } else {
        ohi = __builtin_amdgcn_alignbit(lo, hi, 64 - R);
        olo = __builtin_amdgcn_alignbit(hi, lo, 64 - R);
    }
}

// a ^ (~b & c) as one v_bitop3_b32 (truth table 0xF0 ^ (~0xCC & 0xAA) = 0xD2).  Spelled with the builtin: left to
// pattern matching, the kernels under register pressure (k_shake256_2) got and / not / xor sequences for a fifth of them.
__device__ __forceinline__ uint32_t chi(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xD2);
}

// a ^ b ^ c as one v_bitop3_b32 (truth table 0x96); gfx950 has no v_xor3_b32
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

#define MLDSA_RHOPI(DST, SRC, ROT) rotl64<ROT>(s.lo[SRC] ^ dlo[(SRC) % 5], s.hi[SRC] ^ dhi[(SRC) % 5], blo[DST], bhi[DST])

__device__ __forceinline__ void keccak_round(KeccakState& s, uint32_t rc_lo, uint32_t rc_hi) {
    uint32_t clo[5], chi_[5], dlo[5], dhi[5], blo[25], bhi[25];
#pragma unroll
    for (int x = 0; x < 5; x++) {
        clo[x] = xor3(xor3(s.lo[x], s.lo[x + 5], s.lo[x + 10]), s.lo[x + 15], s.lo[x + 20]);
        chi_[x] = xor3(xor3(s.hi[x], s.hi[x + 5], s.hi[x + 10]), s.hi[x + 15], s.hi[x + 20]);
    }
#pragma unroll
    for (int x = 0; x < 5; x++) {
        uint32_t rl, rh;
        rotl64<1>(clo[(x + 1) % 5], chi_[(x + 1) % 5], rl, rh);
        dlo[x] = clo[(x + 4) % 5] ^ rl;
        dhi[x] = chi_[(x + 4) % 5] ^ rh;
    }
    // rho + pi: B[y][2x+3y] = rotl(A[x][y] ^ D[x], r[x][y]); flat index = x + 5y
    MLDSA_RHOPI(0, 0, 0);
    MLDSA_RHOPI(10, 1, 1);
    MLDSA_RHOPI(20, 2, 62);
    MLDSA_RHOPI(5, 3, 28);
    MLDSA_RHOPI(15, 4, 27);
    MLDSA_RHOPI(16, 5, 36);
    MLDSA_RHOPI(1, 6, 44);
    MLDSA_RHOPI(11, 7, 6);
    MLDSA_RHOPI(21, 8, 55);
    MLDSA_RHOPI(6, 9, 20);
    MLDSA_RHOPI(7, 10, 3);
    MLDSA_RHOPI(17, 11, 10);
    MLDSA_RHOPI(2, 12, 43);
    MLDSA_RHOPI(12, 13, 25);
    MLDSA_RHOPI(22, 14, 39);
    MLDSA_RHOPI(23, 15, 41);
    MLDSA_RHOPI(8, 16, 45);
    MLDSA_RHOPI(18, 17, 15);
    MLDSA_RHOPI(3, 18, 21);
    MLDSA_RHOPI(13, 19, 8);
    MLDSA_RHOPI(14, 20, 18);
    MLDSA_RHOPI(24, 21, 2);
    MLDSA_RHOPI(9, 22, 61);
    MLDSA_RHOPI(19, 23, 56);
    MLDSA_RHOPI(4, 24, 14);
#pragma unroll
    for (int y = 0; y < 25; y += 5) {
#pragma unroll
        for (int x = 0; x < 5; x++) {
            s.lo[y + x] = chi(blo[y + x], blo[y + (x + 1) % 5], blo[y + (x + 2) % 5]);
            s.hi[y + x] = chi(bhi[y + x], bhi[y + (x + 1) % 5], bhi[y + (x + 2) % 5]);
        }
    }
    s.lo[0] ^= rc_lo;
    s.hi[0] ^= rc_hi;
}
#undef MLDSA_RHOPI

__device__ __forceinline__ void keccak_f1600(KeccakState& s) {
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        uint32_t rc_lo = KECCAK_RC_LO[r];
        uint32_t rc_hi = ((KECCAK_RC_HI_BITS >> r) & 1u) << 31;
        keccak_round(s, rc_lo, rc_hi);
    }
}

// little-endian 32-bit load with no alignment requirement (signature / key byte strings
// start at arbitrary byte offsets: SIG_LEN = 3309 and 4627 are odd)
__device__ __forceinline__ uint32_t load_le32(const uint8_t* p) {
    typedef uint32_t __attribute__((aligned(1))) u32_unaligned;  // one byte-granular dword load (gfx950), not four byte loads
    return *reinterpret_cast<const u32_unaligned*>(p);
}

// `bits`-wide little-endian field at bit offset `bo` of an arbitrarily aligned byte string, fetched
// with two ALIGNED dword loads (consecutive lanes read overlapping 8-byte windows: coalesced).
// May read up to 7 bytes past the field: only for fields that are followed by more data.
__device__ __forceinline__ uint32_t load_field_aligned(const uint8_t* base, int bo, int bits) {
    const uintptr_t addr = reinterpret_cast<uintptr_t>(base) + (uintptr_t)(bo >> 3);
    const uint32_t* a4 = reinterpret_cast<const uint32_t*>(addr & ~(uintptr_t)3);
    const int sh = (int)((addr & 3) << 3) + (bo & 7);
    const uint64_t d = ((uint64_t)a4[1] << 32) | a4[0];
    return (uint32_t)(d >> sh) & ((1u << bits) - 1u);
}

// absorb NW 8-byte words from (unaligned) memory into state words 0..NW-1 of a fresh state
template <int NW>
__device__ __forceinline__ void absorb_words(KeccakState& s, const uint8_t* p) {
#pragma unroll
    for (int i = 0; i < NW; i++) {
        s.lo[i] ^= load_le32(p + 8 * i);
        s.hi[i] ^= load_le32(p + 8 * i + 4);
    }
}

// ---- sponge helpers with compile-time word positions (no dynamic register indexing) ----

// XOR an 8-byte little-endian word into lane `w` of the state
__device__ __forceinline__ void xor_word(KeccakState& s, int w, uint32_t lo, uint32_t hi) {
    s.lo[w] ^= lo;
    s.hi[w] ^= hi;
}

// Domain-separation + final pad for a message that ends at byte position `pos` (compile
// time) inside the current block: 0x1F at pos, 0x80 at rate - 1 (FIPS 202 SHAKE).
template <int RATE, int POS>
__device__ __forceinline__ void shake_pad(KeccakState& s) {
    constexpr int w = POS / 8, sh = (POS % 8) * 8;
    if constexpr (sh < 32) s.lo[w] ^= 0x1Fu << sh; else s.hi[w] ^= 0x1Fu << (sh - 32);
    s.hi[RATE / 8 - 1] ^= 0x80000000u;
}

}  // namespace mldsa
