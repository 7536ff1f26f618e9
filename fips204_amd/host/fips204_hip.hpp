// fips204_hip.hpp -- C++ host-side mirror of the integritychain/fips204 API over the C ABI
// (include/mldsa_hip.h).  Header-only; needs nothing but the C ABI (no HIP headers).
//
// Same names, argument meaning and error behaviour as the reference crate:
//   fips204_hip::ml_dsa_44 / ml_dsa_65 / ml_dsa_87        (src/lib.rs:639-740)
//     KG::try_keygen_with_rng / KG::keygen_from_seed       (KeyGen,   src/traits.rs:8-114)
//     PrivateKey::try_sign_with_rng / try_sign_with_seed   (Signer,   src/traits.rs:118-308)
//     PrivateKey::get_public_key
//     PublicKey::verify                                    (Verifier, src/traits.rs:330-362)
//     {PublicKey,PrivateKey}::try_from_bytes / into_bytes  (SerDes,   src/traits.rs:372-424)
//     _internal_sign / _internal_verify                    (src/lib.rs:586-612)
// plus the batched calls the GPU path exists for: keygen_many / sign_many / verify_many.
// Errors: the reference returns Result<_, &'static str>; here a failed Result is a thrown
// fips204_hip::Error carrying the same kind of static message.  verify() never throws on a bad
// signature or an over-long ctx: it returns false (src/lib.rs:368-370).  There is no CPU fallback.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mldsa_hip.h"

namespace fips204_hip {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline void check(int rc, const char* what) {
    if (rc != MLDSA_OK) throw Error(std::string(what) + ": " + mldsa_last_error());
}

// RAII device buffer
class DevBuf {
  public:
    DevBuf() = default;
    explicit DevBuf(size_t bytes) : n_(bytes) { check(mldsa_malloc(&p_, bytes ? bytes : 1), "mldsa_malloc"); }
    DevBuf(const void* host, size_t bytes) : DevBuf(bytes) { upload(host, bytes); }
    DevBuf(DevBuf&& o) noexcept : p_(o.p_), n_(o.n_) { o.p_ = nullptr; o.n_ = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p_ = o.p_; n_ = o.n_; o.p_ = nullptr; o.n_ = 0; } return *this; }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void upload(const void* host, size_t bytes) {
        check(mldsa_memcpy_h2d(p_, host, bytes, nullptr), "mldsa_memcpy_h2d");
        check(mldsa_stream_sync(nullptr), "sync");
    }
    void download(void* host, size_t bytes) const {
        check(mldsa_memcpy_d2h(host, p_, bytes, nullptr), "mldsa_memcpy_d2h");
        check(mldsa_stream_sync(nullptr), "sync");
    }
    template <class T> T* as() const { return static_cast<T*>(p_); }
    size_t size() const { return n_; }
  private:
    void release() { if (p_) { (void)mldsa_memset(p_, 0, n_, nullptr); (void)mldsa_stream_sync(nullptr); (void)mldsa_free(p_); p_ = nullptr; } }
    void* p_ = nullptr;
    size_t n_ = 0;
};

// one context per device, shared by every key object of the process
class Device {
  public:
    static Device& get(int device_id = 0) {
        static Device d(device_id);
        return d;
    }
    mldsa_ctx* ctx() const { return ctx_; }
  private:
    explicit Device(int id) { check(mldsa_ctx_create(id, &ctx_), "mldsa_ctx_create"); }
    ~Device() { mldsa_ctx_destroy(ctx_); }
    mldsa_ctx* ctx_ = nullptr;
};

// concatenate byte strings + u64 offsets (the msgs / ctxs arguments of the C ABI)
struct Packed {
    DevBuf bytes, offsets;
    explicit Packed(const std::vector<std::vector<uint8_t>>& items) {
        std::vector<uint64_t> off(items.size() + 1, 0);
        std::vector<uint8_t> flat;
        for (size_t i = 0; i < items.size(); i++) {
            flat.insert(flat.end(), items[i].begin(), items[i].end());
            off[i + 1] = flat.size();
        }
        if (flat.empty()) flat.push_back(0);
        bytes = DevBuf(flat.data(), flat.size());
        offsets = DevBuf(off.data(), off.size() * sizeof(uint64_t));
    }
};

template <int SET, int K_, int L_, size_t PK, size_t SK, size_t SIG>
struct ParamSet {
    static constexpr int SET_ID = SET, K = K_, L = L_;
    static constexpr size_t PK_LEN = PK, SK_LEN = SK, SIG_LEN = SIG;
    using PkBytes = std::array<uint8_t, PK>;
    using SkBytes = std::array<uint8_t, SK>;
    using Signature = std::array<uint8_t, SIG>;

    // ---- expanded keys on the device, field by field (src/types.rs:19-41) -----------------
    class PublicKeys {  // n keys
      public:
        static PublicKeys try_from_bytes(const std::vector<PkBytes>& pk) {  // expand_public, ml_dsa.rs:477
            PublicKeys k;
            k.n = pk.size();
            k.bytes = pk;
            DevBuf raw(pk.data(), pk.size() * PK);
            k.rho = DevBuf(k.n * 32); k.tr = DevBuf(k.n * 64); k.t1 = DevBuf(k.n * (size_t)K * 1024);
            check(mldsa_pk_expand(Device::get().ctx(), SET, raw.as<uint8_t>(), k.rho.as<uint8_t>(), k.tr.as<uint8_t>(),
                                  k.t1.as<int32_t>(), k.n, nullptr), "mldsa_pk_expand");
            check(mldsa_stream_sync(nullptr), "sync");
            return k;
        }
        size_t n = 0;
        std::vector<PkBytes> bytes;
        DevBuf rho, tr, t1;
    };
    class PrivateKeys {
      public:
        static PrivateKeys try_from_bytes(const std::vector<SkBytes>& sk) {  // expand_private, ml_dsa.rs:445
            PrivateKeys k;
            k.n = sk.size();
            k.bytes = sk;
            DevBuf raw(sk.data(), sk.size() * SK);
            k.rho = DevBuf(k.n * 32); k.cap_k = DevBuf(k.n * 32); k.tr = DevBuf(k.n * 64);
            k.s1 = DevBuf(k.n * (size_t)L * 1024); k.s2 = DevBuf(k.n * (size_t)K * 1024); k.t0 = DevBuf(k.n * (size_t)K * 1024);
            check(mldsa_sk_expand(Device::get().ctx(), SET, raw.as<uint8_t>(), k.rho.as<uint8_t>(), k.cap_k.as<uint8_t>(),
                                  k.tr.as<uint8_t>(), k.s1.as<int32_t>(), k.s2.as<int32_t>(), k.t0.as<int32_t>(), k.n, nullptr),
                  "mldsa_sk_expand");
            check(mldsa_stream_sync(nullptr), "sync");
            return k;
        }
        size_t n = 0;
        std::vector<SkBytes> bytes;
        DevBuf rho, cap_k, tr, s1, s2, t0;
    };

    // ---- batched operations ------------------------------------------------------------------
    static std::pair<std::vector<PkBytes>, std::vector<SkBytes>> keygen_many(const std::vector<std::array<uint8_t, 32>>& xi) {
        const size_t n = xi.size();
        DevBuf dxi(xi.data(), n * 32), dpk(n * PK), dsk(n * SK);
        check(mldsa_keygen(Device::get().ctx(), SET, dxi.as<uint8_t>(), dpk.as<uint8_t>(), dsk.as<uint8_t>(), n, nullptr), "mldsa_keygen");
        std::vector<PkBytes> pk(n);
        std::vector<SkBytes> sk(n);
        dpk.download(pk.data(), n * PK);
        dsk.download(sk.data(), n * SK);
        return {std::move(pk), std::move(sk)};
    }

    static std::vector<Signature> sign_many(const PrivateKeys& sks, const std::vector<uint32_t>& key_idx,
                                            const std::vector<std::vector<uint8_t>>& msgs,
                                            const std::vector<std::vector<uint8_t>>& ctxs,
                                            const std::vector<std::array<uint8_t, 32>>& rnd, int mode = MLDSA_MODE_PURE) {
        const size_t n = msgs.size();
        if (ctxs.size() != n || rnd.size() != n || key_idx.size() != n) throw Error("sign_many: argument lengths differ");
        Packed m(msgs), c(ctxs);
        DevBuf dk(key_idx.data(), n * 4), drnd(rnd.data(), n * 32), dsig(n * SIG), dstat(n * 4);
        check(mldsa_sign(Device::get().ctx(), SET, mode, sks.rho.template as<uint8_t>(), sks.cap_k.template as<uint8_t>(),
                         sks.tr.template as<uint8_t>(), sks.s1.template as<int32_t>(), sks.s2.template as<int32_t>(),
                         sks.t0.template as<int32_t>(), dk.as<uint32_t>(), m.bytes.as<uint8_t>(), m.offsets.as<uint64_t>(),
                         c.bytes.as<uint8_t>(), c.offsets.as<uint64_t>(), drnd.as<uint8_t>(), dsig.as<uint8_t>(),
                         dstat.as<int32_t>(), n, nullptr), "mldsa_sign");
        std::vector<int32_t> st(n);
        dstat.download(st.data(), n * 4);
        for (int32_t s : st)
            if (s == MLDSA_ERR_CTX_LEN) throw Error("ML-DSA.Sign: ctx too long");  // src/lib.rs:274
        std::vector<Signature> sig(n);
        dsig.download(sig.data(), n * SIG);
        return sig;
    }

    static std::vector<bool> verify_many(const PublicKeys& pks, const std::vector<uint32_t>& key_idx,
                                         const std::vector<std::vector<uint8_t>>& msgs, const std::vector<Signature>& sigs,
                                         const std::vector<std::vector<uint8_t>>& ctxs, int mode = MLDSA_MODE_PURE) {
        const size_t n = msgs.size();
        if (ctxs.size() != n || sigs.size() != n || key_idx.size() != n) throw Error("verify_many: argument lengths differ");
        Packed m(msgs), c(ctxs);
        DevBuf dk(key_idx.data(), n * 4), dsig(sigs.data(), n * SIG), dok(n);
        check(mldsa_verify(Device::get().ctx(), SET, mode, pks.rho.template as<uint8_t>(), pks.tr.template as<uint8_t>(),
                           pks.t1.template as<int32_t>(), dk.as<uint32_t>(), m.bytes.as<uint8_t>(), m.offsets.as<uint64_t>(),
                           c.bytes.as<uint8_t>(), c.offsets.as<uint64_t>(), dsig.as<uint8_t>(), dok.as<uint8_t>(), n, nullptr),
              "mldsa_verify");
        std::vector<uint8_t> ok(n);
        dok.download(ok.data(), n);
        return std::vector<bool>(ok.begin(), ok.end());
    }

    // ---- single-key objects with the reference's method names -------------------------------------
    class PublicKey {
      public:
        static PublicKey try_from_bytes(const PkBytes& pk) { return PublicKey(PublicKeys::try_from_bytes({pk})); }
        PkBytes into_bytes() const { return keys_->bytes[0]; }
        // Verifier::verify (src/lib.rs:364-380)
        bool verify(const std::vector<uint8_t>& message, const Signature& sig, const std::vector<uint8_t>& ctx) const {
            return verify_many(*keys_, {0u}, {message}, {sig}, {ctx}, MLDSA_MODE_PURE)[0];
        }
        // _internal_verify (src/lib.rs:605-612)
        bool _internal_verify(const std::vector<uint8_t>& message, const Signature& sig, const std::vector<uint8_t>& ctx) const {
            return verify_many(*keys_, {0u}, {message}, {sig}, {ctx}, MLDSA_MODE_INTERNAL)[0];
        }
      private:
        explicit PublicKey(PublicKeys k) : keys_(std::make_shared<PublicKeys>(std::move(k))) {}
        std::shared_ptr<PublicKeys> keys_;
    };

    class PrivateKey {
      public:
        static PrivateKey try_from_bytes(const SkBytes& sk) { return PrivateKey(PrivateKeys::try_from_bytes({sk})); }
        SkBytes into_bytes() const { return keys_->bytes[0]; }
        // Signer::try_sign_with_seed (src/traits.rs): rnd supplied by the caller
        Signature try_sign_with_seed(const std::array<uint8_t, 32>& rnd, const std::vector<uint8_t>& message,
                                     const std::vector<uint8_t>& ctx) const {
            return sign_many(*keys_, {0u}, {message}, {ctx}, {rnd}, MLDSA_MODE_PURE)[0];
        }
        // Signer::try_sign_with_rng (src/lib.rs:268-296): rnd <- rng (32 bytes)
        template <class Rng>
        Signature try_sign_with_rng(Rng& rng, const std::vector<uint8_t>& message, const std::vector<uint8_t>& ctx) const {
            if (ctx.size() > 255) throw Error("ML-DSA.Sign: ctx too long");  // checked before the rng is touched, lib.rs:274
            std::array<uint8_t, 32> rnd{};
            if (!rng.try_fill_bytes(rnd.data(), 32)) throw Error("ML-DSA.Sign: random number generator failed");
            return try_sign_with_seed(rnd, message, ctx);
        }
        // _internal_sign (src/lib.rs:586-600)
        Signature _internal_sign(const std::vector<uint8_t>& message, const std::vector<uint8_t>& ctx,
                                 const std::array<uint8_t, 32>& rnd) const {
            return sign_many(*keys_, {0u}, {message}, {ctx}, {rnd}, MLDSA_MODE_INTERNAL)[0];
        }
      private:
        explicit PrivateKey(PrivateKeys k) : keys_(std::make_shared<PrivateKeys>(std::move(k))) {}
        std::shared_ptr<PrivateKeys> keys_;
    };

    struct KG {
        // KeyGen::keygen_from_seed (src/lib.rs:247-250)
        static std::pair<PublicKey, PrivateKey> keygen_from_seed(const std::array<uint8_t, 32>& xi) {
            auto ks = keygen_many({xi});
            return {PublicKey::try_from_bytes(ks.first[0]), PrivateKey::try_from_bytes(ks.second[0])};
        }
        // KeyGen::try_keygen_with_rng (src/lib.rs:241-245): xi <- rng (32 bytes)
        template <class Rng>
        static std::pair<PublicKey, PrivateKey> try_keygen_with_rng(Rng& rng) {
            std::array<uint8_t, 32> xi{};
            if (!rng.try_fill_bytes(xi.data(), 32)) throw Error("KeyGen: Random number generator failed");  // ml_dsa.rs:41
            return keygen_from_seed(xi);
        }
    };
};

using ml_dsa_44 = ParamSet<MLDSA_44, 4, 4, 1312, 2560, 2420>;
using ml_dsa_65 = ParamSet<MLDSA_65, 6, 5, 1952, 4032, 3309>;
using ml_dsa_87 = ParamSet<MLDSA_87, 8, 7, 2592, 4896, 4627>;

}  // namespace fips204_hip
