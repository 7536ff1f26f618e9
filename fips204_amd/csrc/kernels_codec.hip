// Wire-format codecs, rounding and the fixed-shape SHAKE256 hashes either side of the hot
// path, so that whole verify / sign / keygen stay on
// the device.  Replaces, for batches: src/conversion.rs (bit_pack / bit_unpack /
// hint_bit_pack / hint_bit_unpack), src/encodings.rs (pk / sk / sig codecs, w1_encode),
// src/high_low.rs (power2round, decompose, high/low bits, make_hint, use_hint) and the
// h256_xof call sites of src/ml_dsa.rs (mu, rho'', c_tilde, tr, keygen seed expansion).
#include <type_traits>

#include <cstdlib>
#include "ctx.h"
#include "challenge_dev.h"
#include "sampler_dev.h"
#include "keccak.h"
#include "keccak_coop2.h"
#include "ntt_wave.h"
#include "rounding.h"
#include "verify_dev.h"

namespace mldsa {

constexpr int CWAVES = 4;
constexpr int CBLOCK = 64 * CWAVES;

template <int I, int E, class F>
__device__ __forceinline__ void static_for_c(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for_c<I + 1, E>(f);
    }
}


// ------------------------------------------------------------------------------------
// Whole-verify arithmetic in one kernel (ml_dsa.rs:368-372, 407-428), ONE WAVE PER OPERATION (the
// structure of k_verify_arith: no barriers, lane-private LDS rows, next A_hat row requested under the
// current row's inverse NTT):
//   decode:  the hint section of the signature -> bit masks in LDS (hint_unpack_wave above)
//   forward: z[j] is unpacked straight from the signature bytes into NTT registers (bit_unpack,
//            conversion.rs:227-262; with the ||z||inf test of ml_dsa.rs:434), then c; results to LDS
//   rows:    A_hat[i] o z_hat - c_hat o t1_hat[i] -> inverse NTT -> UseHint (high_low.rs:155-192)
//            -> w1Encode (encodings.rs:338-360) -> packed bytes
// so neither z, w' nor w1' ever exist as int32 polynomials in HBM.
constexpr int VW = 4;  // waves (= ops in flight) per block
template <int K, int L, int GB, bool G2HI, int MINW, bool APACK>
__global__ __launch_bounds__(64 * VW) __attribute__((amdgpu_waves_per_eu(MINW))) void k_verify_main(
    const int32_t* __restrict__ a_hat, const uint8_t* __restrict__ sigs, size_t sig_len, int ctilde_len,
    const int32_t* __restrict__ c, const int32_t* __restrict__ t1, const uint32_t* __restrict__ key_idx,
    int32_t* __restrict__ hvalid, int omega, uint8_t* __restrict__ w1, size_t w1_stride, int32_t* __restrict__ znorm,
    int32_t zbound, size_t n_ops, const Twiddle* __restrict__ fwd_tab, const Twiddle* __restrict__ inv_tab, int a_by_key) {
    constexpr int CB = GB + 1;
    constexpr int BITS = G2HI ? 4 : 6;
    __shared__ int4 zh[VW][L + 1][64];
    __shared__ uint32_t hint_lds[VW][HINT_LDS_DWORDS];
    __shared__ Twiddle tw_lds[(FWD_TW + INV_TW) * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < FWD_TW * 64; i += 64 * VW) tw_lds[i] = fwd_tab[i];
    for (int i = threadIdx.x; i < INV_TW * 64; i += 64 * VW) tw_lds[FWD_TW * 64 + i] = inv_tab[i];
    __syncthreads();
    const LdsTw ftw{tw_lds, lane};
    const LdsTw itw{tw_lds + FWD_TW * 64, lane};
    const uint32_t wid = blockIdx.x * VW + wave, n_waves = gridDim.x * VW;

    for (size_t op = wid; op < n_ops; op += n_waves) {
        // the lane index made opaque per operation: address arithmetic on it is then redone where it is used instead of being
        // hoisted out of the op loop and kept (or, at the register budget of 4 waves per SIMD, spilled) across it: 18-24 VGPRs
        // less (ML-DSA-87: 120, no spill at 4 waves per SIMD), -6 % / -3 % run time for ML-DSA-65 / 87 in a same-box A/B.
        // (The same trick made k_verify_arith ~4 % slower and left k_sign_tail unchanged: not applied there.)
        unsigned ul = (unsigned)lane;
        asm volatile("" : "+v"(ul));
        const size_t key = key_idx ? (size_t)__builtin_amdgcn_readfirstlane((int)key_idx[op]) : op;
        const size_t aop = a_by_key ? key : op;  // per-key A_hat kept by the caller, or the op's own ExpandA output
        // APACK: A_hat in the pipelines' 24-bit form (768 bytes per polynomial, three dwords per lane)
        using ARow = std::conditional_t<APACK, Packed3, int4>;
        const ARow* arow = APACK ? reinterpret_cast<const ARow*>(reinterpret_cast<const uint32_t*>(a_hat) + (aop * K * (size_t)L) * PACKED_POLY_DWORDS)
                                 : reinterpret_cast<const ARow*>(a_hat + (aop * K * (size_t)L) * N);
        auto coeffs = [](const ARow& v) -> int4 {
            if constexpr (APACK) return unpack24(v); else return v;
        };
        const int4* trow = reinterpret_cast<const int4*>(t1 + (key * K) * (size_t)N);
        // (nontemporal, field.h "Cache policy": the op's own ExpandA output is read exactly once -- APACK; a per-key A_hat kept by the caller is
        //  shared by the ops of that key and stays on the default policy)
        ARow av[L];
#pragma unroll
        for (int j = 0; j < L; j++) av[j] = load_row<NT_A_VERIFY && APACK>(&arow[(unsigned)(j * 64) + ul]);
        int4 tv = trow[ul];
        // ---- (c_tilde, z, h) <- sigDecode: z in the forward loop, the hint bytes requested here and decoded after it
        const uint8_t* zsrc = sigs + op * sig_len + ctilde_len;
        const HintBytes hbytes = hint_load(zsrc + L * (32 * CB), omega, K, (int)ul);
        bool zbad = false;
#pragma unroll 1
        for (int j = 0; j <= L; j++) {
            asm volatile("" ::: "memory");  // keep the LDS twiddle reads at their point of use
            int32_t r[4];
            if (j < L) {
                const uint8_t* src = zsrc + (size_t)j * (32 * CB);
                int32_t mx = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    // one byte-granular dword load per field (field.h; the same encoding as the signer's raw y); hints follow z:
                    // the over-read is in-buffer
                    r[k] = y_from_raw<CB>(y_raw_dword<CB, NT_ZC>(src, k, (int)ul), (int)ul);
                    const int32_t a = r[k] < 0 ? -r[k] : r[k];
                    mx = a > mx ? a : mx;
                }
                zbad |= mx >= zbound;
            } else {  // c: k_sample_in_ball<.., C8>'s bytes, the lane's four coefficients in its dword
                const uint32_t d = load_once<NT_ZC>(reinterpret_cast<const uint32_t*>(c) + op * 64 + ul);
                r[0] = (int8_t)(d & 0xFF); r[1] = (int8_t)((d >> 8) & 0xFF); r[2] = (int8_t)((d >> 16) & 0xFF); r[3] = (int8_t)(d >> 24);
            }
            ntt_fwd_wave(r, ftw, lane);
            if (j == L) {
#pragma unroll
                for (int k = 0; k < 4; k++) r[k] = mont_mul(r[k], 1);  // c_hat * 2^-32
            }
            zh[wave][j][lane] = make_int4(r[0], r[1], r[2], r[3]);
        }
        const bool hint_ok = hint_unpack_wave<K>(hbytes, omega, hint_lds[wave], (int)ul);  // the masks stay in LDS for the rows below
        // ||z||inf >= gamma1 - beta (ml_dsa.rs:434) and a malformed hint section as per-op flags for the verdict
        if (lane == 0) {
            znorm[op] = __ballot(zbad) != 0ull ? 0x7fffffff : 0;
            hvalid[op] = hint_ok ? 1 : 0;
        }
        // ---- rows
#pragma unroll 1
        for (int i = 0; i < K; i++) {
            asm volatile("" ::: "memory");
            // 64-bit accumulation, one Montgomery reduction per coefficient and row (field.h): |a| < 2^24 (what ExpandA
            // produces), |z_hat| < 9 q, at most L + 1 <= 8 terms: |sum| < 2^54 = the reduction's input bound
            int64_t acc64[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < L; j++) {
                const int4 zv = zh[wave][j][lane];
                const int4 a4 = coeffs(av[j]);
                acc64[0] += (int64_t)a4.x * zv.x;
                acc64[1] += (int64_t)a4.y * zv.y;
                acc64[2] += (int64_t)a4.z * zv.z;
                acc64[3] += (int64_t)a4.w * zv.w;
            }
            const int4 cv = zh[wave][L][lane];  // c_hat * 2^-32 in (-q, q); t1 from mldsa_pk_expand is in (-q, q)
            acc64[0] -= (int64_t)cv.x * tv.x;
            acc64[1] -= (int64_t)cv.y * tv.y;
            acc64[2] -= (int64_t)cv.z * tv.z;
            acc64[3] -= (int64_t)cv.w * tv.w;
            int32_t acc[4];
            if (i + 1 < K) {  // next row: in flight during this row's inverse transform
#pragma unroll
                for (int j = 0; j < L; j++) av[j] = load_row<NT_A_VERIFY && APACK>(&arow[(unsigned)(((i + 1) * L + j) * 64) + ul]);
                tv = trow[(unsigned)((i + 1) * 64) + ul];
            }
            // hint bits of coefficients 64 k + lane: mask word 2 k + (lane >> 5).  One dword per lane (lane l holds word l & 7),
            // handed out with v_readlane after the inverse transform: one live register instead of four
            const uint32_t hword = hint_lds[wave][24 + i * 8 + (ul & 7)];
#pragma unroll
            for (int k = 0; k < 4; k++) acc[k] = mont_reduce64(acc64[k]);  // (-q, q): the inverse transform's input range
            ntt_inv_wave(acc, itw, lane, F_MONT2);
            // acc[k] = w'[64 k + lane], canonical.  UseHint, then pack BITS-bit fields.
            uint8_t* dst = w1 + op * w1_stride + (size_t)i * (32 * BITS);
            uint32_t v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)hword, 2 * k), whi = (uint32_t)__builtin_amdgcn_readlane((int)hword, 2 * k + 1);
                const uint32_t hwk = lane < 32 ? wlo : whi;
                const uint32_t h = (hwk >> (lane & 31)) & 1u;
                v[k] = (uint32_t)use_hint<G2HI>((int32_t)h, acc[k]);
            }
            pack_w1_strided<G2HI>(v, dst, lane);
        }
    }
}

// ------------------------------------------------------------------------------------
// mu = H(tr || M', 64) (ml_dsa.rs:185-196 / 386-397), one op per lane, variable-length
// messages.  M' = M (internal), 0x00 | len(ctx) | ctx | M (pure) or 0x01 | len(ctx) | ctx |
// OID | PH(M) (pre-hash; the caller passes OID | PH(M) as the message).  Each lane builds
// its 136-byte rate block in a lane-private LDS row, then absorbs it.
constexpr int MU_BLK_STRIDE = 35;  // dwords per lane row (136 bytes + pad, odd stride)

// Offsets are the caller's and are never trusted: the call vouches for the bytes [off[0], off[n_call]) of msgs / ctxs, and an op
// is hashed only if its pair lies inside that range in order (off[0] <= off[i] <= off[i + 1] <= off[n_call]).  Any other pair
// -- decreasing, wrapping, pointing past the end -- refuses the op (flag 2: ok = 0 / MLDSA_ERR_PARAM) without reading a byte of
// it; a monotonic table never trips this.  An op whose ctx is longer than 255 bytes is refused before anything of it is read,
// like the reference's early return (lib.rs:274, 368): a 100 MB ctx costs what an empty one costs.
// msg_off / ctx_off: the CALL's tables (n_call + 1 entries); this launch covers ops [op0, op0 + n_ops) of it, and the per-op
// arrays (tr without key_idx, mu, ctx_bad, key_bad, key_idx) are the launch's own (index op - op0).
__global__ __launch_bounds__(64) void k_mu(const uint8_t* __restrict__ tr, size_t tr_stride,
                                           const uint32_t* __restrict__ key_idx, int mode,
                                           const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ msg_off,
                                           const uint8_t* __restrict__ ctxs, const uint64_t* __restrict__ ctx_off,
                                           uint8_t* __restrict__ mu, size_t mu_stride, int32_t* __restrict__ ctx_bad,
                                           const int32_t* __restrict__ key_bad, size_t n_ops, size_t op0, size_t n_call) {
    __shared__ uint32_t blk[64 * MU_BLK_STRIDE];
    const int lane = threadIdx.x;
    const size_t op = (size_t)blockIdx.x * 64 + lane;
    const bool valid = op < n_ops;
    uint32_t* row = blk + lane * MU_BLK_STRIDE;
    uint8_t* rowb = reinterpret_cast<uint8_t*>(row);

    const uint8_t *trp = nullptr, *mp = nullptr, *cp = nullptr;
    size_t mlen = 0, clen = 0;
    bool live = false;  // the op is hashed
    if (valid) {
        trp = tr + (key_idx ? key_idx[op] : op) * tr_stride;
        const uint64_t m0 = msg_off[op0 + op], m1 = msg_off[op0 + op + 1];
        bool bad_off = !(msg_off[0] <= m0 && m0 <= m1 && m1 <= msg_off[n_call]);
        mp = msgs + m0;
        mlen = (size_t)(m1 - m0);
        bad_off |= mlen != 0 && msgs == nullptr;  // offsets that name bytes of a NULL array
        if (ctx_off) {
            const uint64_t c0 = ctx_off[op0 + op], c1 = ctx_off[op0 + op + 1];
            bad_off |= !(ctx_off[0] <= c0 && c0 <= c1 && c1 <= ctx_off[n_call]);
            cp = ctxs + c0;
            clen = (size_t)(c1 - c0);
            bad_off |= clen != 0 && ctxs == nullptr;
        }
        // 2: malformed offsets or key index out of range; 1: ctx too long (lib.rs:274, 368, 589, 605: every entry point)
        const int flag = bad_off ? 2 : clen > 255 ? 1 : (key_bad ? key_bad[op] : 0);
        if (ctx_bad) ctx_bad[op] = flag;
        live = !bad_off && clen <= 255;
        if (!live) mlen = clen = 0;
    }
    const size_t pre = (mode == MLDSA_MODE_INTERNAL) ? 0 : 2 + clen;
    const size_t total = live ? 64 + pre + mlen : 0;
    const size_t my_blocks = live ? total / SHAKE256_RATE + 1 : 0;  // the pad always fits in the last block
    size_t max_blocks = my_blocks;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const size_t o = (size_t)__shfl_xor((unsigned long long)max_blocks, m);
        max_blocks = o > max_blocks ? o : max_blocks;
    }
    KeccakState st;
    keccak_zero(st);
    for (size_t b = 0; b < max_blocks; b++) {
        if (b < my_blocks) {
            const size_t base = b * SHAKE256_RATE;
            auto byte_at = [&](size_t pos) -> uint8_t {
                if (pos < total) {
                    if (pos < 64) return trp[pos];
                    if (pos < 64 + pre) {
                        const size_t q = pos - 64;
                        return q == 0 ? (uint8_t)(mode == MLDSA_MODE_PREHASH ? 1 : 0) : q == 1 ? (uint8_t)clen : cp[q - 2];
                    }
                    return mp[pos - 64 - pre];
                }
                return pos == total ? (uint8_t)0x1F : (uint8_t)0;
            };
            // dword by dword: whole dwords of tr and of the message come from one (byte-granular) load each, only the dwords
            // that straddle a boundary (prefix, message end, pad) are assembled from bytes
            for (int i = 0; i < SHAKE256_RATE / 4; i++) {
                const size_t pos = base + 4 * (size_t)i;
                uint32_t v;
                if (pos + 4 <= 64) v = load_le32(trp + pos);
                else if (pos >= 64 + pre && pos + 4 <= total) v = load_le32(mp + (pos - 64 - pre));
                else if (pos > total) v = 0;
                else v = (uint32_t)byte_at(pos) | ((uint32_t)byte_at(pos + 1) << 8) | ((uint32_t)byte_at(pos + 2) << 16) | ((uint32_t)byte_at(pos + 3) << 24);
                row[i] = v;
            }
            if (b == my_blocks - 1) rowb[SHAKE256_RATE - 1] |= 0x80;
            static_for_c<0, 17>([&](auto wc) {
                constexpr int W = decltype(wc)::value;
                st.lo[W] ^= row[2 * W];
                st.hi[W] ^= row[2 * W + 1];
            });
            keccak_f1600(st);
        }
    }
    if (valid) {
        uint32_t* out = reinterpret_cast<uint32_t*>(mu + op * mu_stride);
#pragma unroll
        for (int i = 0; i < 8; i++) { out[2 * i] = st.lo[i]; out[2 * i + 1] = st.hi[i]; }
    }
}

// ------------------------------------------------------------------------------------
// Fixed-shape SHAKE256 over two concatenated device buffers A (la bytes, per-op stride sa,
// optional index) and B (lb bytes) plus up to 4 literal tail bytes: out = first OUT bytes.
//   c_tilde' = H(mu | w1)  (ml_dsa.rs:233, 429)     rho'' = H(K | rnd | mu)  (ml_dsa.rs:199)
//   tr = H(pk)             (ml_dsa.rs:100, 486)     keygen seed = H(xi | K | L, 128) (ml_dsa.rs:68)
// One op per lane for the permutation, but the INPUT is loaded cooperatively: for every rate
// block the wave reads the 64 ops' 136-byte pieces with consecutive lanes on consecutive
// dwords (coalesced) into an LDS tile, and each lane then absorbs its own row.  la and lb are
// multiples of 4; ALIGNED = all pointers / strides are multiples of 4 (dword loads).

// With vd.ok != nullptr the kernel is the tail of verify_internal (ml_dsa.rs:429-436): instead of storing the digest it
// compares it with the signature's c_tilde and writes the verdict, combined with the decode failures that make the
// reference return false early (ml_dsa.rs:368-376, lib.rs:368-370).
struct VerdictArgs {
    const uint8_t* sigs;
    size_t sig_len;
    const int32_t* znorm;
    int32_t zbound;
    const int32_t* hvalid;
    const int32_t* ctx_bad;
    uint8_t* ok;
};

template <int OUT, bool ALIGNED>
__global__ __launch_bounds__(CBLOCK) void k_shake256_2(const uint8_t* __restrict__ a, size_t sa, int la,
                                                       const uint32_t* __restrict__ a_idx,
                                                       const uint8_t* __restrict__ b, size_t sb, int lb,
                                                       uint32_t tail, int tail_len,
                                                       uint8_t* __restrict__ out, size_t so, size_t n_ops,
                                                       const uint32_t* __restrict__ n_dev, VerdictArgs vd,
                                                       const uint32_t* __restrict__ b_idx) {
    __shared__ uint32_t tiles[CWAVES * 64 * H_STRIDE];
    __shared__ unsigned long long ptr_a[CWAVES * 64], ptr_b[CWAVES * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t* tile = tiles + wave * 64 * H_STRIDE;
    if (n_dev) n_ops = *n_dev;  // the signer's rounds: the count lives on the device, the grid is sized from its expectation
    if constexpr (EXP_PRIO != 0) __builtin_amdgcn_s_setprio(EXP_PRIO);  // (experiment: the latency chain ahead of the streaming kernels beside it)
    for (size_t base_op = (size_t)blockIdx.x * CBLOCK; base_op < n_ops; base_op += (size_t)gridDim.x * CBLOCK) {
        const size_t op = base_op + threadIdx.x;
        const bool valid = op < n_ops;
        // rows of ops past the end of the batch point at the tile's first op: their loads stay legal, their results unused
        const size_t opc = valid ? op : base_op;
        const unsigned long long pa = (unsigned long long)(a + (a_idx ? a_idx[opc] : opc) * sa);
        ptr_a[wave * 64 + lane] = pa;
        ptr_b[wave * 64 + lane] = b ? (unsigned long long)(b + (b_idx ? b_idx[opc] : opc) * sb) : pa;
        wave_lds_sync_c();
        KeccakState st;
        shake256_2_absorb<ALIGNED>(st, tile, ptr_a + wave * 64, ptr_b + wave * 64, la, lb, tail, tail_len, lane);
        if (valid && vd.ok) {
            const uint8_t* c0 = vd.sigs + op * vd.sig_len;  // c_tilde opens the signature (encodings.rs:251)
            uint32_t diff = 0;
            static_for_c<0, OUT / 8>([&](auto wc) {
                constexpr int W = decltype(wc)::value;
                diff |= (st.lo[W] ^ load_le32(c0 + 8 * W)) | (st.hi[W] ^ load_le32(c0 + 8 * W + 4));
            });
            vd.ok[op] = (uint8_t)(diff == 0 && vd.znorm[op] < vd.zbound && vd.hvalid[op] && !vd.ctx_bad[op]);
        } else if (valid) {
            uint8_t* po = out + op * so;
            static_for_c<0, OUT / 8>([&](auto wc) {
                constexpr int W = decltype(wc)::value;
                typedef uint32_t __attribute__((aligned(1))) u32_unaligned;  // (any alignment: pk / sk rows are odd-sized)
                *reinterpret_cast<u32_unaligned*>(po + 8 * W) = st.lo[W];
                *reinterpret_cast<u32_unaligned*>(po + 8 * W + 4) = st.hi[W];
            });
        }
    }
}

// The same hash for SMALL batches: one state per wave on the interleaved cooperative sponge (keccak_coop2.h), 2.2 instead of 9.4 us per
// permutation.  Same arguments, same results (ALIGNED is not needed: every load is a byte-granular dword load).  The two lanes holding
// a state word fetch its eight bytes of each rate block straight from A | B | tail | pad; the digest leaves through the lanes holding
// words 0 .. OUT / 8 - 1 (the E lane the word's low dword, the O lane the high one).
template <int OUT>
__global__ __launch_bounds__(CBLOCK) void k_shake256_2_coop(const uint8_t* __restrict__ a, size_t sa, int la, const uint32_t* __restrict__ a_idx,
                                                            const uint8_t* __restrict__ b, size_t sb, int lb, uint32_t tail, int tail_len,
                                                            uint8_t* __restrict__ out, size_t so, size_t n_ops, const uint32_t* __restrict__ n_dev,
                                                            VerdictArgs vd, const uint32_t* __restrict__ b_idx) {
    const int lane = threadIdx.x & 63, half = lane >> 5;
    const Coop2Lane c = coop2_lane(lane);
    if (n_dev) n_ops = *n_dev;
    const int total = la + lb + tail_len, blocks = total / SHAKE256_RATE + 1;  // the pad always fits in the last block
    const size_t wave0 = (size_t)blockIdx.x * CWAVES + (threadIdx.x >> 6), stride = (size_t)gridDim.x * CWAVES;
    for (size_t op = wave0; op < n_ops; op += stride) {  // wave-uniform
        const uint8_t* pa = a + (a_idx ? a_idx[op] : op) * sa;
        const uint8_t* pb = b ? b + (b_idx ? b_idx[op] : op) * sb : pa;
        auto msg_dword = [&](int off) -> uint32_t {  // la, lb are multiples of 4: a dword never straddles A | B
            if (off + 4 <= la) return load_le32(pa + off);
            if (off + 4 <= la + lb) return load_le32(pb + (off - la));
            uint32_t v = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int t = off + k - (la + lb);  // position in tail | pad
                const uint32_t byte = t < tail_len ? (tail >> (8 * t)) & 0xFFu : t == tail_len ? 0x1Fu : 0u;
                v |= byte << (8 * k);
            }
            if (off + 4 == blocks * SHAKE256_RATE) v |= 0x80000000u;
            return v;
        };
        uint32_t v = 0;
        const bool absorbs = c.active && c.word < SHAKE256_RATE / 8;
        for (int blk = 0; blk < blocks; blk++) {
            if (absorbs) {
                const int off = blk * SHAKE256_RATE + 8 * c.word;
                v ^= coop2_from_lohi(msg_dword(off), msg_dword(off + 4), c);
            }
            keccak_f1600_coop2(v, c);
        }
        uint32_t lo, hi;
        coop2_to_lohi(v, lane, lo, hi);
        const uint32_t mine = half ? hi : lo;  // this lane's dword of the digest: bytes 8 word + 4 half
        const bool digest = c.active && c.word < OUT / 8;
        if (vd.ok) {
            const uint8_t* c0 = vd.sigs + op * vd.sig_len + 8 * c.word + 4 * half;  // c_tilde opens the signature (encodings.rs:251)
            const bool differs = digest && (mine ^ load_le32(c0)) != 0;
            const unsigned long long any = __ballot(differs);
            if (lane == 0) vd.ok[op] = (uint8_t)(any == 0ull && vd.znorm[op] < vd.zbound && vd.hvalid[op] && !vd.ctx_bad[op]);
        } else if (digest) {
            typedef uint32_t __attribute__((aligned(1))) u32_unaligned;  // (any alignment: pk / sk rows are odd-sized)
            *reinterpret_cast<u32_unaligned*>(out + op * so + 8 * c.word + 4 * half) = mine;
        }
    }
}

// mu = H(tr | M') for SMALL calls: one op per wave on the same sponge (k_mu's checks, flags and bytes; any message length).
__global__ __launch_bounds__(CBLOCK) void k_mu_coop(const uint8_t* __restrict__ tr, size_t tr_stride, const uint32_t* __restrict__ key_idx, int mode,
                                                    const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ msg_off, const uint8_t* __restrict__ ctxs,
                                                    const uint64_t* __restrict__ ctx_off, uint8_t* __restrict__ mu, size_t mu_stride,
                                                    int32_t* __restrict__ ctx_bad, const int32_t* __restrict__ key_bad, size_t n_ops, size_t op0,
                                                    size_t n_call) {
    const int lane = threadIdx.x & 63;
    const Coop2Lane c = coop2_lane(lane);
    const size_t wave0 = (size_t)blockIdx.x * CWAVES + (threadIdx.x >> 6), stride = (size_t)gridDim.x * CWAVES;
    for (size_t op = wave0; op < n_ops; op += stride) {  // wave-uniform
        uint32_t lo, hi;
        const int flag = mu_coop2(tr + (key_idx ? key_idx[op] : op) * tr_stride, mode, msgs, msg_off, ctxs, ctx_off, op0 + op, n_call,
                                  key_bad ? key_bad[op] : 0, lo, hi, lane, c);
        if (c.active && c.word < 8) reinterpret_cast<uint32_t*>(mu + op * mu_stride)[2 * c.word + (lane >> 5)] = lane < 32 ? lo : hi;
        if (lane == 0 && ctx_bad) ctx_bad[op] = flag;
    }
}

// ------------------------------------------------------------------------- launchers
static inline unsigned lane_blocks(size_t n) { return (unsigned)((n + CBLOCK - 1) / CBLOCK); }

int launch_verify_main(mldsa_ctx* ctx, const mldsa_params* p, const int32_t* a_hat, const uint8_t* sigs, const int32_t* c,
                       const int32_t* t1, const uint32_t* key_idx, int32_t* hvalid, uint8_t* w1, size_t w1_stride,
                       int32_t* znorm, size_t n_ops, hipStream_t s, bool a_by_key, bool a_packed) {
    if (n_ops == 0) return MLDSA_OK;
    dim3 grid(grid_for(ctx, n_ops, VW, 16));
#define MLDSA_VM2(KK, LL, GB, G2, MW, AP)                                                                                    \
    hipLaunchKernelGGL((k_verify_main<KK, LL, GB, G2, MW, AP>), grid, dim3(64 * VW), 0, s, a_hat, sigs,                       \
                       (size_t)p->sig_len, p->ctilde_len, c, t1, key_idx, hvalid, p->omega, w1, w1_stride, znorm, p->gamma1 - p->beta, n_ops, \
                       ctx->d_fwd_tw, ctx->d_inv_tw, a_by_key ? 1 : 0)
#define MLDSA_VM(KK, LL, GB, G2, MW) do { if (a_packed) MLDSA_VM2(KK, LL, GB, G2, MW, true); else MLDSA_VM2(KK, LL, GB, G2, MW, false); } while (0)
    if (p->set == MLDSA_44) MLDSA_VM(4, 4, 17, false, 5);
    else if (p->set == MLDSA_65) MLDSA_VM(6, 5, 19, true, 4);
    // ML-DSA-87 with the packed A_hat (21 prefetched dwords per lane): compiled for 3 waves per SIMD (137 VGPRs, no
    // spill); at 4 waves the compiler spills 8 VGPRs of loop-invariant addresses (36 B scratch) for 1.3 % more throughput
    else if (a_packed) MLDSA_VM2(8, 7, 19, true, 4, true);
    else MLDSA_VM2(8, 7, 19, true, 4, false);
#undef MLDSA_VM2
#undef MLDSA_VM
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_mu(mldsa_ctx* ctx, const uint8_t* tr, size_t tr_stride, const uint32_t* key_idx, int mode, const uint8_t* msgs,
              const uint64_t* msg_off, const uint8_t* ctxs, const uint64_t* ctx_off, uint8_t* mu, size_t mu_stride,
              int32_t* ctx_bad, size_t n_ops, hipStream_t s, const int32_t* key_bad, size_t op0, size_t n_call) {
    if (ctx && ctx->opt_coop_hash && n_ops <= ctx->coop_mu_max) {  // a small call: one op per wave, 2.2 us per permutation instead of 9.4
        hipLaunchKernelGGL(k_mu_coop, dim3((unsigned)((n_ops + CWAVES - 1) / CWAVES)), dim3(CBLOCK), 0, s, tr, tr_stride, key_idx, mode, msgs, msg_off,
                           ctxs, ctx_off, mu, mu_stride, ctx_bad, key_bad, n_ops, op0, n_call);
        MLDSA_HIP_CHECK(hipGetLastError());
        return MLDSA_OK;
    }
    hipLaunchKernelGGL(k_mu, dim3((unsigned)((n_ops + 63) / 64)), dim3(64), 0, s, tr, tr_stride, key_idx, mode, msgs, msg_off,
                       ctxs, ctx_off, mu, mu_stride, ctx_bad, key_bad, n_ops, op0, n_call);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

static int launch_shake256_2v(mldsa_ctx* ctx, int out_len, const uint8_t* a, size_t sa, int la, const uint32_t* a_idx, const uint8_t* b,
                              size_t sb, int lb, uint32_t tail, int tail_len, uint8_t* out, size_t so, size_t n_ops, hipStream_t s,
                              const uint32_t* n_dev, const VerdictArgs& vd, const uint32_t* b_idx = nullptr);

int launch_shake256_2(mldsa_ctx* ctx, int out_len, const uint8_t* a, size_t sa, int la, const uint32_t* a_idx, const uint8_t* b,
                      size_t sb, int lb, uint32_t tail, int tail_len, uint8_t* out, size_t so, size_t n_ops, hipStream_t s,
                      const uint32_t* n_dev, const uint32_t* b_idx) {
    VerdictArgs none{};
    return launch_shake256_2v(ctx, out_len, a, sa, la, a_idx, b, sb, lb, tail, tail_len, out, so, n_ops, s, n_dev, none, b_idx);
}

// c_tilde' = H(mu | w1Encode(w1')) and the final verdict of verify_internal in one kernel (ml_dsa.rs:429-436)
int launch_ctilde_verdict(mldsa_ctx* ctx, const mldsa_params* p, const uint8_t* mu_w1, size_t mw, const uint8_t* sigs, const int32_t* znorm,
                          const int32_t* hvalid, const int32_t* ctx_bad, uint8_t* ok, size_t n_ops, hipStream_t s) {
    VerdictArgs vd{sigs, (size_t)p->sig_len, znorm, p->gamma1 - p->beta, hvalid, ctx_bad, ok};
    return launch_shake256_2v(ctx, p->ctilde_len, mu_w1, mw, (int)mw, nullptr, nullptr, 0, 0, 0, 0, nullptr, 0, n_ops, s, nullptr, vd);
}

static int launch_shake256_2v(mldsa_ctx* ctx, int out_len, const uint8_t* a, size_t sa, int la, const uint32_t* a_idx, const uint8_t* b,
                              size_t sb, int lb, uint32_t tail, int tail_len, uint8_t* out, size_t so, size_t n_ops, hipStream_t s,
                              const uint32_t* n_dev, const VerdictArgs& vd, const uint32_t* b_idx) {
    if (n_ops == 0 && !n_dev) return MLDSA_OK;
    dim3 grid(lane_blocks(n_ops ? n_ops : 1)), block(CBLOCK);
    if ((la & 3) != 0 || (lb & 3) != 0) return set_error(MLDSA_ERR_PARAM, "shake256_2: segment lengths must be multiples of 4 bytes");
    // Small batches (for the signer's rounds: small expected row counts): the wave-cooperative form, two ops per wavefront.  Up to
    // 4 096 ops (mldsa_ctx::coop_hash_max) that is at most two waves per SIMD, where it still runs a permutation in 5.7 us against 9.4.
    if (ctx->opt_coop_hash && n_ops <= ctx->coop_hash_max) {
        const dim3 cgrid((unsigned)((std::max<size_t>(n_ops, 1) + CWAVES - 1) / CWAVES));
#define MLDSA_COOP_CASE(O)                                                                                                                           \
    case O: hipLaunchKernelGGL((k_shake256_2_coop<O>), cgrid, block, 0, s, a, sa, la, a_idx, b, sb, lb, tail, tail_len, out, so, n_ops, n_dev, vd, b_idx); break;
        switch (out_len) {
            MLDSA_COOP_CASE(32)
            MLDSA_COOP_CASE(48)
            MLDSA_COOP_CASE(64)
            MLDSA_COOP_CASE(128)
            default: return set_error(MLDSA_ERR_PARAM, "shake256_2: unsupported output length");
        }
#undef MLDSA_COOP_CASE
        MLDSA_HIP_CHECK(hipGetLastError());
        return MLDSA_OK;
    }
    const bool al = (((uintptr_t)a | (uintptr_t)sa | (uintptr_t)b | (uintptr_t)sb) & 3) == 0;
#define MLDSA_SHAKE_CASE(O)                                                                                              \
    case O:                                                                                                              \
        if (al) hipLaunchKernelGGL((k_shake256_2<O, true>), grid, block, 0, s, a, sa, la, a_idx, b, sb, lb, tail, tail_len, out, so, n_ops, n_dev, vd, b_idx); \
        else hipLaunchKernelGGL((k_shake256_2<O, false>), grid, block, 0, s, a, sa, la, a_idx, b, sb, lb, tail, tail_len, out, so, n_ops, n_dev, vd, b_idx);   \
        break;
    switch (out_len) {
        MLDSA_SHAKE_CASE(32)
        MLDSA_SHAKE_CASE(48)
        MLDSA_SHAKE_CASE(64)
        MLDSA_SHAKE_CASE(128)
        default: return set_error(MLDSA_ERR_PARAM, "shake256_2: unsupported output length");
    }
#undef MLDSA_SHAKE_CASE
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

// ------------------------------------------------------------------------------------
// The codecs as seams of their own (SURVEY rows F1 w1Encode, F2 bit_pack / bit_unpack / hint_bit_pack / hint_bit_unpack /
// sig_encode / sig_decode).  Inside the pipelines these are prologues and epilogues of k_verify_main, k_sign_tail and k_resolve;
// the kernels below run the same device helpers (y_from_raw, hint_unpack_wave, pack_w1_strided) on int32 polynomials in HBM so
// that parity tests reach the reference's seams directly.  Not on the timed path.

// bit_pack (conversion.rs:143-186; simple_bit_pack 120-132 = a == 0): one thread per 8 coefficients = `bitlen` bytes
__global__ __launch_bounds__(256) void k_bit_pack(const int32_t* __restrict__ w, int a, int b, int bitlen, uint8_t* __restrict__ out, size_t n_polys) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_polys * 32) return;
    const int4 lo = reinterpret_cast<const int4*>(w)[2 * t], hi = reinterpret_cast<const int4*>(w)[2 * t + 1];
    const int32_t c[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    uint8_t* dst = out + t * (size_t)bitlen;
    uint64_t acc = 0;
    int bits = 0, o = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        // conversion.rs:166-171: b.abs_diff(coeff) for a > 0, coeff.unsigned_abs() otherwise; masked to the field (a coefficient
        // outside [-a, b] is a debug_assert in the reference: here it cannot touch its neighbours)
        const uint32_t f = a > 0 ? (uint32_t)(b > c[i] ? b - c[i] : c[i] - b) : (uint32_t)(c[i] < 0 ? -c[i] : c[i]);
        acc |= (uint64_t)(f & ((1u << bitlen) - 1u)) << bits;
        bits += bitlen;
        while (bits >= 8) { dst[o++] = (uint8_t)acc; acc >>= 8; bits -= 8; }
    }
}

// bit_unpack (conversion.rs:227-262; simple_bit_unpack 198-213 = a == 0); ok[poly] = the reference's Ok / Err (256-261)
__global__ __launch_bounds__(256) void k_bit_unpack(const uint8_t* __restrict__ v, int a, int b, int bitlen, int32_t* __restrict__ w,
                                                    uint8_t* __restrict__ ok, size_t n_polys) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < n_polys * 32;
    bool bad = false;
    if (live) {
        const uint8_t* src = v + t * (size_t)bitlen;
        const int32_t bot = b - (1 << bitlen) + 1;  // -|b - 2^c + 1|: never positive for the reference's (a, b) pairs
        uint64_t acc = 0;
        int bits = 0, o = 0;
        int32_t c[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            while (bits < bitlen) { acc |= (uint64_t)src[o++] << bits; bits += 8; }
            const int32_t f = (int32_t)(acc & ((1u << bitlen) - 1u));
            acc >>= bitlen;
            bits -= bitlen;
            c[i] = a == 0 ? f : b - f;
            bad |= c[i] < (bot < 0 ? bot : -bot) || c[i] > b;
        }
        reinterpret_cast<int4*>(w)[2 * t] = make_int4(c[0], c[1], c[2], c[3]);
        reinterpret_cast<int4*>(w)[2 * t + 1] = make_int4(c[4], c[5], c[6], c[7]);
    }
    // 32 threads per polynomial = half a wave
    const unsigned long long m = __ballot(bad);
    const int lane = threadIdx.x & 63;
    if (live && ok && (lane & 31) == 0) ok[t >> 5] = ((m >> (lane & 32)) & 0xFFFFFFFFull) == 0 ? 1 : 0;
}

// hint_bit_pack (conversion.rs:277-328) by one wave: positions in coefficient order through ballots.  ok = 0 when the weight
// exceeds omega (the reference's debug_assert at 286-289; positions past omega are dropped, never written over the limits)
template <int K>
__device__ __forceinline__ bool hint_pack_wave(const int32_t* __restrict__ h, int omega, uint8_t* __restrict__ y, int lane) {
    for (int i = lane; i < omega + K; i += 64) y[i] = 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int index = 0;
#pragma unroll 1
    for (int i = 0; i < K; i++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool bit = h[i * N + 64 * k + lane] != 0;
            const unsigned long long mask = __ballot(bit);
            if (bit) {
                const int rank = index + __popcll(mask & ((1ull << lane) - 1ull));
                if (rank < omega) y[rank] = (uint8_t)(64 * k + lane);
            }
            index += __popcll(mask);
        }
        if (lane == 0) y[omega + i] = (uint8_t)(index < omega ? index : omega);
    }
    return index <= omega;
}

// the masks hint_unpack_wave left in LDS -> K int32 polynomials of 0 / 1 (all zero for a refused hint section)
template <int K>
__device__ __forceinline__ void hint_masks_to_polys(const uint32_t* __restrict__ hl, bool valid, int32_t* __restrict__ h, int lane) {
#pragma unroll 1
    for (int i = 0; i < K; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = 64 * k + lane;
            h[i * N + j] = valid ? (int32_t)((hl[24 + i * 8 + (j >> 5)] >> (j & 31)) & 1u) : 0;
        }
}

template <int K>
__global__ __launch_bounds__(CBLOCK) void k_hint_pack(const int32_t* __restrict__ h, int omega, uint8_t* __restrict__ y, uint8_t* __restrict__ ok,
                                                      size_t n_ops) {
    const int lane = threadIdx.x & 63;
    const size_t op = (size_t)blockIdx.x * CWAVES + (threadIdx.x >> 6);
    if (op >= n_ops) return;
    const bool good = hint_pack_wave<K>(h + op * K * (size_t)N, omega, y + op * (size_t)(omega + K), lane);
    if (ok && lane == 0) ok[op] = good ? 1 : 0;
}

template <int K>
__global__ __launch_bounds__(CBLOCK) void k_hint_unpack(const uint8_t* __restrict__ y, int omega, int32_t* __restrict__ h, uint8_t* __restrict__ ok,
                                                        size_t n_ops) {
    __shared__ uint32_t hint_lds[CWAVES][HINT_LDS_DWORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t op = (size_t)blockIdx.x * CWAVES + wave;
    if (op >= n_ops) return;
    const HintBytes hb = hint_load(y + op * (size_t)(omega + K), omega, K, lane);
    const bool good = hint_unpack_wave<K>(hb, omega, hint_lds[wave], lane);
    hint_masks_to_polys<K>(hint_lds[wave], good, h + op * K * (size_t)N, lane);
    if (ok && lane == 0) ok[op] = good ? 1 : 0;
}

// sig_decode (encodings.rs:290-328): c_tilde | z (bit_unpack with gamma1 - 1, gamma1: the forward loop of k_verify_main) | h
template <int K, int L, int CB>
__global__ __launch_bounds__(CBLOCK) void k_sig_decode(const uint8_t* __restrict__ sigs, size_t sig_len, int ctilde_len, int omega,
                                                       uint8_t* __restrict__ c_tilde, int32_t* __restrict__ z, int32_t* __restrict__ h,
                                                       uint8_t* __restrict__ ok, size_t n_ops) {
    __shared__ uint32_t hint_lds[CWAVES][HINT_LDS_DWORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t op = (size_t)blockIdx.x * CWAVES + wave;
    if (op >= n_ops) return;
    const uint8_t* sig = sigs + op * sig_len;
    if (lane < ctilde_len) c_tilde[op * (size_t)ctilde_len + lane] = sig[lane];
    const uint8_t* zsrc = sig + ctilde_len;
    const HintBytes hb = hint_load(zsrc + L * (32 * CB), omega, K, lane);
#pragma unroll 1
    for (int j = 0; j < L; j++)
#pragma unroll
        for (int k = 0; k < 4; k++)
            z[(op * L + j) * (size_t)N + 64 * k + lane] = y_from_raw<CB>(y_raw_dword<CB>(zsrc + (size_t)j * (32 * CB), k, lane), lane);
    const bool good = hint_unpack_wave<K>(hb, omega, hint_lds[wave], lane);
    hint_masks_to_polys<K>(hint_lds[wave], good, h + op * K * (size_t)N, lane);
    if (ok && lane == 0) ok[op] = good ? 1 : 0;
}

// sig_encode (encodings.rs:238-280): z as in the signer's tail (four consecutive coefficients = 9 / 10 bytes per lane)
template <int K, int L, int CB>
__global__ __launch_bounds__(CBLOCK) void k_sig_encode(const uint8_t* __restrict__ c_tilde, const int32_t* __restrict__ z, const int32_t* __restrict__ h,
                                                       size_t sig_len, int ctilde_len, int omega, uint8_t* __restrict__ sigs,
                                                       uint8_t* __restrict__ ok, size_t n_ops) {
    const int lane = threadIdx.x & 63;
    const size_t op = (size_t)blockIdx.x * CWAVES + (threadIdx.x >> 6);
    if (op >= n_ops) return;
    uint8_t* sig = sigs + op * sig_len;
    if (lane < ctilde_len) sig[lane] = c_tilde[op * (size_t)ctilde_len + lane];
    constexpr int32_t GAMMA1 = 1 << (CB - 1);
    bool bad = false;
#pragma unroll 1
    for (int j = 0; j < L; j++) {
        const int4 z4 = reinterpret_cast<const int4*>(z + (op * L + j) * (size_t)N)[lane];
        const int32_t zz[4] = {z4.x, z4.y, z4.z, z4.w};
        uint64_t lo = 0;
        uint32_t hi = 0;
#pragma unroll
        for (int t = 0; t < 4; t++) {
            bad |= zz[t] <= -GAMMA1 || zz[t] > GAMMA1;  // the reference's debug_assert (encodings.rs:249)
            const uint64_t f = (uint64_t)((uint32_t)(GAMMA1 - zz[t]) & ((1u << CB) - 1u));
            const int sh = t * CB;
            lo |= f << sh;
            if (sh + CB > 64) hi |= (uint32_t)(f >> (64 - sh));
        }
        constexpr int NBYTES = CB / 2;
        uint8_t* dst = sig + ctilde_len + (size_t)j * (32 * CB) + (size_t)lane * NBYTES;
        *reinterpret_cast<u64_any*>(dst) = lo;
        if constexpr (NBYTES == 10) *reinterpret_cast<u16_any*>(dst + 8) = (uint16_t)hi;
        else dst[8] = (uint8_t)hi;
    }
    const bool hint_ok = hint_pack_wave<K>(h + op * K * (size_t)N, omega, sig + ctilde_len + (size_t)L * (32 * CB), lane);
    const bool z_ok = __ballot(bad) == 0ull;
    if (ok && lane == 0) ok[op] = hint_ok && z_ok ? 1 : 0;
}

// w1_encode (encodings.rs:338-360): one wave per polynomial, the packing k_verify_main and the signer's w kernel use
template <bool G2HI>
__global__ __launch_bounds__(CBLOCK) void k_w1_encode(const int32_t* __restrict__ w1, uint8_t* __restrict__ out, size_t n_polys) {
    constexpr int BITS = G2HI ? 4 : 6;
    const int lane = threadIdx.x & 63;
    const size_t poly = (size_t)blockIdx.x * CWAVES + (threadIdx.x >> 6);
    if (poly >= n_polys) return;
    uint32_t v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = (uint32_t)w1[poly * N + 64 * k + lane] & ((1u << BITS) - 1u);
    pack_w1_strided<G2HI>(v, out + poly * (size_t)(32 * BITS), lane);
}

static inline unsigned wave_blocks(size_t n) { return (unsigned)((n + CWAVES - 1) / CWAVES); }

int launch_bit_pack(mldsa_ctx*, const int32_t* w, int a, int b, int bitlen, uint8_t* out, size_t n_polys, hipStream_t s) {
    if (n_polys == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_bit_pack, dim3((unsigned)((n_polys * 32 + 255) / 256)), dim3(256), 0, s, w, a, b, bitlen, out, n_polys);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_bit_unpack(mldsa_ctx*, const uint8_t* v, int a, int b, int bitlen, int32_t* w, uint8_t* ok, size_t n_polys, hipStream_t s) {
    if (n_polys == 0) return MLDSA_OK;
    hipLaunchKernelGGL(k_bit_unpack, dim3((unsigned)((n_polys * 32 + 255) / 256)), dim3(256), 0, s, v, a, b, bitlen, w, ok, n_polys);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

#define MLDSA_BY_SET(p, KERNEL3, ...)                                                                               \
    do {                                                                                                            \
        if ((p)->set == MLDSA_44) hipLaunchKernelGGL((KERNEL3<4, 4, 18>), __VA_ARGS__);                             \
        else if ((p)->set == MLDSA_65) hipLaunchKernelGGL((KERNEL3<6, 5, 20>), __VA_ARGS__);                        \
        else hipLaunchKernelGGL((KERNEL3<8, 7, 20>), __VA_ARGS__);                                                  \
    } while (0)
#define MLDSA_BY_K(p, KERNEL1, ...)                                                                                 \
    do {                                                                                                            \
        if ((p)->k == 4) hipLaunchKernelGGL((KERNEL1<4>), __VA_ARGS__);                                             \
        else if ((p)->k == 6) hipLaunchKernelGGL((KERNEL1<6>), __VA_ARGS__);                                        \
        else hipLaunchKernelGGL((KERNEL1<8>), __VA_ARGS__);                                                         \
    } while (0)

int launch_hint_pack(mldsa_ctx*, const mldsa_params* p, const int32_t* h, uint8_t* y, uint8_t* ok, size_t n_ops, hipStream_t s) {
    if (n_ops == 0) return MLDSA_OK;
    MLDSA_BY_K(p, k_hint_pack, dim3(wave_blocks(n_ops)), dim3(CBLOCK), 0, s, h, p->omega, y, ok, n_ops);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_hint_unpack(mldsa_ctx*, const mldsa_params* p, const uint8_t* y, int32_t* h, uint8_t* ok, size_t n_ops, hipStream_t s) {
    if (n_ops == 0) return MLDSA_OK;
    MLDSA_BY_K(p, k_hint_unpack, dim3(wave_blocks(n_ops)), dim3(CBLOCK), 0, s, y, p->omega, h, ok, n_ops);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_sig_decode(mldsa_ctx*, const mldsa_params* p, const uint8_t* sigs, uint8_t* c_tilde, int32_t* z, int32_t* h, uint8_t* ok, size_t n_ops,
                      hipStream_t s) {
    if (n_ops == 0) return MLDSA_OK;
    MLDSA_BY_SET(p, k_sig_decode, dim3(wave_blocks(n_ops)), dim3(CBLOCK), 0, s, sigs, (size_t)p->sig_len, p->ctilde_len, p->omega, c_tilde, z, h, ok,
                 n_ops);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_sig_encode(mldsa_ctx*, const mldsa_params* p, const uint8_t* c_tilde, const int32_t* z, const int32_t* h, uint8_t* sigs, uint8_t* ok,
                      size_t n_ops, hipStream_t s) {
    if (n_ops == 0) return MLDSA_OK;
    MLDSA_BY_SET(p, k_sig_encode, dim3(wave_blocks(n_ops)), dim3(CBLOCK), 0, s, c_tilde, z, h, (size_t)p->sig_len, p->ctilde_len, p->omega, sigs, ok,
                 n_ops);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

int launch_w1_encode(mldsa_ctx*, const mldsa_params* p, const int32_t* w1, uint8_t* out, size_t n_ops, hipStream_t s) {
    const size_t n_polys = n_ops * p->k;
    if (n_polys == 0) return MLDSA_OK;
    if (p->gamma2 == (Q - 1) / 32) hipLaunchKernelGGL(k_w1_encode<true>, dim3(wave_blocks(n_polys)), dim3(CBLOCK), 0, s, w1, out, n_polys);
    else hipLaunchKernelGGL(k_w1_encode<false>, dim3(wave_blocks(n_polys)), dim3(CBLOCK), 0, s, w1, out, n_polys);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}
#undef MLDSA_BY_SET
#undef MLDSA_BY_K

// h256_xof / g128_xof (hashing.rs:13-27) as a seam: SHAKE256 / SHAKE128 of one byte string per op (the reference's list of slices,
// concatenated by the caller), any input length, the first out_len bytes of the output.  One op per lane like every sponge of the
// pipelines (keccak.h); the rate block is assembled in a lane-private LDS row.  Offsets are validated like k_mu's: an op whose
// pair is malformed is not read, gets a zero output and bad[op] = 1.  Not on the timed path (the pipelines' hashes have fixed
// shapes: k_mu, k_shake256_2, the samplers).
template <int RATE>
__global__ __launch_bounds__(64) void k_xof(const uint8_t* __restrict__ data, const uint64_t* __restrict__ off, uint8_t* __restrict__ out,
                                            size_t out_len, uint8_t* __restrict__ bad, size_t n_ops) {
    constexpr int RW = RATE / 4;       // dwords per block
    constexpr int STRIDE = RW | 1;     // odd row stride: lane rows on distinct banks
    __shared__ uint32_t blk[64 * STRIDE];
    const int lane = threadIdx.x;
    const size_t op = (size_t)blockIdx.x * 64 + lane;
    const bool valid = op < n_ops;
    uint32_t* row = blk + lane * STRIDE;
    const uint8_t* src = nullptr;
    size_t len = 0;
    bool live = false;
    if (valid) {
        const uint64_t a = off[op], b = off[op + 1];
        const bool bad_off = !(off[0] <= a && a <= b && b <= off[n_ops]) || (b != a && data == nullptr);
        if (bad) bad[op] = bad_off ? 1 : 0;
        live = !bad_off;
        if (live) { src = data + a; len = (size_t)(b - a); }
    }
    const size_t in_blocks = live ? len / RATE + 1 : 0;  // the pad always fits in the last block
    const size_t out_blocks = live ? (out_len + RATE - 1) / RATE : 0;
    KeccakState st;
    keccak_zero(st);
    for (size_t b = 0; __ballot(b < in_blocks) != 0ull; b++) {
        if (b < in_blocks) {
            const size_t base = b * RATE;
            for (int i = 0; i < RW; i++) {
                const size_t pos = base + 4 * (size_t)i;
                uint32_t v = 0;
                if (pos + 4 <= len) v = load_le32(src + pos);
                else
                    for (int t = 0; t < 4; t++) {
                        const size_t q = pos + t;
                        v |= (uint32_t)(q < len ? src[q] : q == len ? 0x1Fu : 0u) << (8 * t);
                    }
                row[i] = v;
            }
            if (b == in_blocks - 1) row[RW - 1] ^= 0x80000000u;
            static_for_c<0, RATE / 8>([&](auto wc) {
                constexpr int W = decltype(wc)::value;
                st.lo[W] ^= row[2 * W];
                st.hi[W] ^= row[2 * W + 1];
            });
            keccak_f1600(st);
        }
    }
    uint8_t* dst = out + op * out_len;
    for (size_t b = 0; __ballot(b < out_blocks) != 0ull; b++) {
        if (b < out_blocks) {
            if (b) keccak_f1600(st);
            static_for_c<0, RATE / 8>([&](auto wc) {
                constexpr int W = decltype(wc)::value;
                row[2 * W] = st.lo[W];
                row[2 * W + 1] = st.hi[W];
            });
            const size_t base = b * RATE, nb = out_len - base < (size_t)RATE ? out_len - base : (size_t)RATE;
            const uint8_t* rb = reinterpret_cast<const uint8_t*>(row);
            for (size_t i = 0; i < nb; i++) dst[base + i] = rb[i];
        }
    }
    if (valid && !live)
        for (size_t i = 0; i < out_len; i++) dst[i] = 0;
}

int launch_xof(mldsa_ctx*, int bits, const uint8_t* data, const uint64_t* off, uint8_t* out, size_t out_len, uint8_t* bad, size_t n_ops, hipStream_t s) {
    if (n_ops == 0 || out_len == 0) return MLDSA_OK;
    const dim3 grid((unsigned)((n_ops + 63) / 64)), block(64);
    if (bits == 128) hipLaunchKernelGGL(k_xof<SHAKE128_RATE>, grid, block, 0, s, data, off, out, out_len, bad, n_ops);
    else hipLaunchKernelGGL(k_xof<SHAKE256_RATE>, grid, block, 0, s, data, off, out, out_len, bad, n_ops);
    MLDSA_HIP_CHECK(hipGetLastError());
    return MLDSA_OK;
}

}  // namespace mldsa
