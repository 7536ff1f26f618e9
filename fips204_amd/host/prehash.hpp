// HashML-DSA message front-end on the host: Ph (src/types.rs:5-12) and hash_message (src/hashing.rs:316-354).
//
// The pre-hash is message-length-bound host work: the device receives OID || PH(M) as the
// message of a MLDSA_MODE_PREHASH call and builds M' = 0x01 | len(ctx) | ctx | OID | PH(M) inside k_mu
// (src/ml_dsa.rs:192-194).  The reference takes SHA-256 / SHA-512 / SHAKE128 from the sha2 / sha3 crates; this header
// carries plain FIPS 180-4 / FIPS 202 implementations so that the C++ mirror has no dependency beyond libmldsa_hip.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <vector>

namespace fips204_hip {

enum class Ph { SHA256, SHA512, SHAKE128 };

namespace prehash_detail {

template <class W> inline W rotr(W x, int n) { return (W)((x >> n) | (x << (8 * (int)sizeof(W) - n))); }

// ---- SHA-256 / SHA-512 (FIPS 180-4): the round constants are the fractional parts of the cube roots of the first
// primes, the initial values those of the square roots; both are derived here once instead of being listed.
inline bool is_prime(unsigned n) {
    for (unsigned d = 2; d * d <= n; d++)
        if (n % d == 0) return false;
    return n >= 2;
}

// floor(frac(p^(1/k)) * 2^64) for k = 2, 3: the largest f with (ip * 2^64 + f)^k <= p * 2^(64 k), decided bit by bit
// with exact multi-limb integer arithmetic (32-bit limbs, the numbers stay below 2^224)
inline uint64_t frac_root(unsigned p, int k) {
    using Big = std::array<uint32_t, 8>;
    auto mul = [](const Big& x, const Big& y) {
        Big r{};
        for (int i = 0; i < 8; i++) {
            uint64_t carry = 0;
            for (int j = 0; i + j < 8; j++) {
                const uint64_t t = (uint64_t)x[i] * y[j] + r[i + j] + carry;
                r[i + j] = (uint32_t)t;
                carry = t >> 32;
            }
        }
        return r;
    };
    auto le = [](const Big& x, const Big& y) {
        for (int i = 7; i >= 0; i--)
            if (x[i] != y[i]) return x[i] < y[i];
        return true;
    };
    unsigned ip = 1;
    while (true) {
        unsigned long long v = 1;
        for (int i = 0; i < k; i++) v *= (ip + 1);
        if (v > p) break;
        ip++;
    }
    Big bound{};
    bound[2 * k] = p;  // p * 2^(64 k)
    uint64_t f = 0;
    for (int b = 63; b >= 0; b--) {
        const uint64_t t = f | (1ull << b);
        const Big a{(uint32_t)t, (uint32_t)(t >> 32), ip, 0, 0, 0, 0, 0};
        Big r = a;
        for (int i = 1; i < k; i++) r = mul(r, a);
        if (le(r, bound)) f = t;
    }
    return f;
}

struct Sha2Tables {
    uint32_t k256[64], h256[8];
    uint64_t k512[80], h512[8];
    Sha2Tables() {
        unsigned p = 2;
        for (int i = 0; i < 80; p++) {
            if (!is_prime(p)) continue;
            k512[i] = frac_root(p, 3);
            if (i < 64) k256[i] = (uint32_t)(k512[i] >> 32);
            if (i < 8) {
                h512[i] = frac_root(p, 2);
                h256[i] = (uint32_t)(h512[i] >> 32);
            }
            i++;
        }
    }
};
inline const Sha2Tables& sha2_tables() {
    static const Sha2Tables t;
    return t;
}

template <class W, int ROUNDS, int S0a, int S0b, int S0c, int S1a, int S1b, int S1c, int s0a, int s0b, int s0c, int s1a, int s1b, int s1c>
inline void sha2_compress(W* h, const uint8_t* block, const W* k) {
    W w[ROUNDS];
    for (int i = 0; i < 16; i++) {
        W v = 0;
        for (size_t b = 0; b < sizeof(W); b++) v = (W)((v << 8) | block[i * sizeof(W) + b]);
        w[i] = v;
    }
    for (int i = 16; i < ROUNDS; i++) {
        const W a = w[i - 15], b = w[i - 2];
        w[i] = w[i - 16] + (rotr(a, s0a) ^ rotr(a, s0b) ^ (a >> s0c)) + w[i - 7] + (rotr(b, s1a) ^ rotr(b, s1b) ^ (b >> s1c));
    }
    W s[8];
    for (int i = 0; i < 8; i++) s[i] = h[i];
    for (int i = 0; i < ROUNDS; i++) {
        const W t1 = s[7] + (rotr(s[4], S1a) ^ rotr(s[4], S1b) ^ rotr(s[4], S1c)) + ((s[4] & s[5]) ^ (~s[4] & s[6])) + k[i] + w[i];
        const W t2 = (rotr(s[0], S0a) ^ rotr(s[0], S0b) ^ rotr(s[0], S0c)) + ((s[0] & s[1]) ^ (s[0] & s[2]) ^ (s[1] & s[2]));
        for (int j = 7; j > 0; j--) s[j] = s[j - 1];
        s[4] += t1;
        s[0] = t1 + t2;
    }
    for (int i = 0; i < 8; i++) h[i] += s[i];
}

template <class W, class Compress>
inline void sha2_run(W* h, const uint8_t* msg, size_t len, uint8_t* out, size_t out_len, Compress compress) {
    constexpr size_t B = 16 * sizeof(W);
    size_t i = 0;
    for (; i + B <= len; i += B) compress(h, msg + i);
    uint8_t tail[2 * B] = {0};
    const size_t rem = len - i;
    if (rem) std::memcpy(tail, msg + i, rem);
    tail[rem] = 0x80;
    const size_t total = (rem + 1 + 2 * sizeof(W) <= B) ? B : 2 * B;  // length field: 2 words (the high one stays 0)
    const uint64_t bits = (uint64_t)len * 8;
    for (int b = 0; b < 8; b++) tail[total - 1 - b] = (uint8_t)(bits >> (8 * b));
    compress(h, tail);
    if (total == 2 * B) compress(h, tail + B);
    for (size_t j = 0; j < out_len; j++) out[j] = (uint8_t)(h[j / sizeof(W)] >> (8 * (sizeof(W) - 1 - j % sizeof(W))));
}

// ---- SHAKE128 (FIPS 202), host-side Keccak-f[1600]
struct KeccakTables {
    int rot[25], dst[25];
    uint64_t rc[24];
    // rotation offsets by the (x, y) -> (y, 2x + 3y) walk, pi as a destination index, round constants by the LFSR
    KeccakTables() {
        int x = 1, y = 0;
        rot[0] = 0;
        for (int t = 0; t < 24; t++) {
            rot[x + 5 * y] = ((t + 1) * (t + 2) / 2) % 64;
            const int nx = y, ny = (2 * x + 3 * y) % 5;
            x = nx;
            y = ny;
        }
        for (int xx = 0; xx < 5; xx++)
            for (int yy = 0; yy < 5; yy++) dst[xx + 5 * yy] = yy + 5 * ((2 * xx + 3 * yy) % 5);
        uint8_t lfsr = 1;
        for (int r = 0; r < 24; r++) {
            uint64_t c = 0;
            for (int j = 0; j < 7; j++) {
                if (lfsr & 1) c |= 1ull << ((1 << j) - 1);
                lfsr = (uint8_t)((lfsr << 1) ^ ((lfsr & 0x80) ? 0x71 : 0));
            }
            rc[r] = c;
        }
    }
};

inline void keccak_f1600(uint64_t a[25]) {
    static const KeccakTables tab;
    const int *rot = tab.rot, *dst = tab.dst;
    const uint64_t* rc = tab.rc;
    for (int r = 0; r < 24; r++) {
        uint64_t c[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) {
            const uint64_t d = c[(x + 4) % 5] ^ ((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63));
            for (int y = 0; y < 25; y += 5) a[x + y] ^= d;
        }
        for (int i = 0; i < 25; i++) b[dst[i]] = rot[i] ? ((a[i] << rot[i]) | (a[i] >> (64 - rot[i]))) : a[i];
        for (int y = 0; y < 25; y += 5)
            for (int x = 0; x < 5; x++) a[x + y] = b[x + y] ^ (~b[(x + 1) % 5 + y] & b[(x + 2) % 5 + y]);
        a[0] ^= rc[r];
    }
}

inline void shake128(const uint8_t* msg, size_t len, uint8_t* out, size_t out_len) {
    constexpr size_t RATE = 168;
    uint64_t a[25] = {0};
    auto xor_byte = [&](size_t pos, uint8_t v) { a[pos / 8] ^= (uint64_t)v << (8 * (pos % 8)); };
    size_t pos = 0;
    for (size_t i = 0; i < len; i++) {
        xor_byte(pos++, msg[i]);
        if (pos == RATE) {
            keccak_f1600(a);
            pos = 0;
        }
    }
    xor_byte(pos, 0x1F);
    xor_byte(RATE - 1, 0x80);
    for (size_t o = 0; o < out_len; o++) {
        if (o % RATE == 0) keccak_f1600(a);
        out[o] = (uint8_t)(a[(o % RATE) / 8] >> (8 * (o % 8)));
    }
}

}  // namespace prehash_detail

inline std::array<uint8_t, 32> sha256(const uint8_t* msg, size_t len) {
    using namespace prehash_detail;
    const Sha2Tables& t = sha2_tables();
    uint32_t h[8];
    std::memcpy(h, t.h256, sizeof h);
    std::array<uint8_t, 32> out;
    sha2_run<uint32_t>(h, msg, len, out.data(), 32, [&](uint32_t* hh, const uint8_t* b) {
        sha2_compress<uint32_t, 64, 2, 13, 22, 6, 11, 25, 7, 18, 3, 17, 19, 10>(hh, b, t.k256);
    });
    return out;
}

inline std::array<uint8_t, 64> sha512(const uint8_t* msg, size_t len) {
    using namespace prehash_detail;
    const Sha2Tables& t = sha2_tables();
    uint64_t h[8];
    std::memcpy(h, t.h512, sizeof h);
    std::array<uint8_t, 64> out;
    sha2_run<uint64_t>(h, msg, len, out.data(), 64, [&](uint64_t* hh, const uint8_t* b) {
        sha2_compress<uint64_t, 80, 28, 34, 39, 14, 18, 41, 1, 8, 7, 19, 61, 6>(hh, b, t.k512);
    });
    return out;
}

// hash_message (src/hashing.rs:316-354): the 11-byte DER OID of the hash followed by PH(M) (32 / 64 / 32 bytes)
inline std::vector<uint8_t> hash_message(const std::vector<uint8_t>& message, Ph ph) {
    static const uint8_t oid[11] = {0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02, 0x00};
    std::vector<uint8_t> out;
    out.reserve(11 + 64);
    out.assign(oid, oid + 11);
    switch (ph) {
    case Ph::SHA256: {
        out[10] = 0x01;
        const auto d = sha256(message.data(), message.size());
        out.insert(out.end(), d.begin(), d.end());
        break;
    }
    case Ph::SHA512: {
        out[10] = 0x03;
        const auto d = sha512(message.data(), message.size());
        out.insert(out.end(), d.begin(), d.end());
        break;
    }
    case Ph::SHAKE128: {
        out[10] = 0x0B;
        uint8_t d[32];
        prehash_detail::shake128(message.data(), message.size(), d, 32);
        out.insert(out.end(), d, d + 32);
        break;
    }
    }
    return out;
}

}  // namespace fips204_hip
