// fips204_hip.hpp -- C++ host-side mirror of the integritychain/fips204 API over the C ABI
// (include/mldsa_hip.h).  Header-only; needs nothing but the C ABI (no HIP headers).
//
// Same names, argument meaning and error behaviour as the reference crate:
//   fips204_hip::ml_dsa_44 / ml_dsa_65 / ml_dsa_87        (src/lib.rs:639-740)
//     KG::try_keygen / try_keygen_with_rng / keygen_from_seed        (KeyGen, src/traits.rs:8-114)
//     PrivateKey::try_sign / try_sign_with_rng / try_sign_with_seed  (Signer, src/traits.rs:118-308)
//     PrivateKey::get_public_key
//     PublicKey::verify                                    (Verifier, src/traits.rs:330-362)
//     PrivateKey::try_hash_sign_with_rng / _with_seed, PublicKey::hash_verify with Ph::{SHA256,SHA512,SHAKE128}
//                                                          (HashML-DSA, src/lib.rs:310-342, 391-411; prehash.hpp)
//     {PublicKey,PrivateKey}::try_from_bytes / into_bytes  (SerDes,   src/traits.rs:372-424)
//     _internal_sign / _internal_verify                    (src/lib.rs:586-612)
// plus the batched calls the GPU path exists for: keygen_many / sign_many / verify_many (device-resident
// expanded keys) and keygen_host / sign_host / verify_host (everything in host memory, wire-format keys: the
// library stages sub-batches itself with upload, kernels and download overlapped).
// Errors: the reference returns Result<_, &'static str>; here a failed Result is a thrown
// fips204_hip::Error carrying the same kind of static message.  verify() never throws on a bad
// signature or an over-long ctx: it returns false (src/lib.rs:368-370).  There is no CPU fallback.
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mldsa_hip.h"
#include "prehash.hpp"

namespace fips204_hip {

// rand_core::OsRng, the generator behind the reference's try_keygen / try_sign / try_hash_sign (src/traits.rs:45, 157, 250):
// the kernel's CSPRNG through /dev/urandom
struct OsRng {
    bool try_fill_bytes(uint8_t* out, size_t n) {
        std::FILE* f = std::fopen("/dev/urandom", "rb");
        if (!f) return false;
        const size_t got = std::fread(out, 1, n, f);
        std::fclose(f);
        return got == n;
    }
};

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

inline void check(int rc, const char* what) {
    if (rc != MLDSA_OK) throw Error(std::string(what) + ": " + mldsa_last_error());
}

// one context per device, created on first use and shared by every key object of the process.  The
// library binds the calling thread to the context's device inside every call, so objects of several
// devices can be used side by side; Device::use(id) picks the device new key objects are created on.
class Device {
  public:
    static Device& get(int device_id = -1) {
        static std::mutex mu;
        static std::map<int, std::unique_ptr<Device>> devices;
        std::lock_guard<std::mutex> lk(mu);
        if (device_id < 0) device_id = current();
        auto& d = devices[device_id];
        if (!d) d.reset(new Device(device_id));
        return *d;
    }
    static void use(int device_id) { current() = device_id; }
    mldsa_ctx* ctx() const { return ctx_; }
    int id() const { return id_; }
    ~Device() { mldsa_ctx_destroy(ctx_); }
  private:
    static int& current() { static thread_local int cur = 0; return cur; }
    explicit Device(int id) : id_(id) { check(mldsa_ctx_create(id, &ctx_), "mldsa_ctx_create"); }
    mldsa_ctx* ctx_ = nullptr;
    int id_ = 0;
};

// RAII device buffer on the current Device
class DevBuf {
  public:
    DevBuf() = default;
    explicit DevBuf(size_t bytes) : n_(bytes) { check(mldsa_ctx_malloc(Device::get().ctx(), &p_, bytes ? bytes : 1), "mldsa_ctx_malloc"); }
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

// first byte of a vector of fixed-size byte arrays (std::array is contiguous and has no header); nullptr for an empty vector --
// `v.data()->data()` would be a member call on a null pointer there (found by the UBSan run of tests/cpp/test_mirror_asan.cpp)
template <class A>
inline const uint8_t* raw_bytes(const std::vector<A>& v) { return reinterpret_cast<const uint8_t*>(v.data()); }
template <class A>
inline uint8_t* raw_bytes(std::vector<A>& v) { return reinterpret_cast<uint8_t*>(v.data()); }

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
            DevBuf raw(pk.data(), pk.size() * PK);
            k.rho = DevBuf(k.n * 32); k.tr = DevBuf(k.n * 64); k.t1 = DevBuf(k.n * (size_t)K * 1024);
            check(mldsa_pk_expand(Device::get().ctx(), SET, raw.as<uint8_t>(), k.rho.as<uint8_t>(), k.tr.as<uint8_t>(),
                                  k.t1.as<int32_t>(), k.n, nullptr), "mldsa_pk_expand");
            check(mldsa_stream_sync(nullptr), "sync");
            return k;
        }
        // SerDes::into_bytes (src/lib.rs:478-493): recomputed from the expanded fields on the device
        std::vector<PkBytes> into_bytes() const {
            DevBuf out(n * PK);
            check(mldsa_pk_into_bytes(Device::get().ctx(), SET, rho.template as<uint8_t>(), t1.template as<int32_t>(), out.as<uint8_t>(), n,
                                      nullptr), "mldsa_pk_into_bytes");
            std::vector<PkBytes> pk(n);
            out.download(pk.data(), n * PK);
            return pk;
        }
        size_t n = 0;
        DevBuf rho, tr, t1;
    };
    class PrivateKeys {
      public:
        static PrivateKeys try_from_bytes(const std::vector<SkBytes>& sk) {  // expand_private, ml_dsa.rs:445
            PrivateKeys k;
            k.n = sk.size();
            DevBuf raw(sk.data(), sk.size() * SK);
            k.rho = DevBuf(k.n * 32); k.cap_k = DevBuf(k.n * 32); k.tr = DevBuf(k.n * 64);
            k.s1 = DevBuf(k.n * (size_t)L * 1024); k.s2 = DevBuf(k.n * (size_t)K * 1024); k.t0 = DevBuf(k.n * (size_t)K * 1024);
            check(mldsa_sk_expand(Device::get().ctx(), SET, raw.as<uint8_t>(), k.rho.as<uint8_t>(), k.cap_k.as<uint8_t>(),
                                  k.tr.as<uint8_t>(), k.s1.as<int32_t>(), k.s2.as<int32_t>(), k.t0.as<int32_t>(), k.n, nullptr),
                  "mldsa_sk_expand");
            check(mldsa_stream_sync(nullptr), "sync");
            return k;
        }
        // SerDes::into_bytes (src/lib.rs:427-465)
        std::vector<SkBytes> into_bytes() const {
            DevBuf out(n * SK);
            check(mldsa_sk_into_bytes(Device::get().ctx(), SET, rho.template as<uint8_t>(), cap_k.template as<uint8_t>(),
                                      tr.template as<uint8_t>(), s1.template as<int32_t>(), s2.template as<int32_t>(),
                                      t0.template as<int32_t>(), out.as<uint8_t>(), n, nullptr), "mldsa_sk_into_bytes");
            std::vector<SkBytes> sk(n);
            out.download(sk.data(), n * SK);
            return sk;
        }
        // Signer::get_public_key (src/lib.rs:345-349 -> private_to_public_key, src/ml_dsa.rs:502-559)
        PublicKeys get_public_key() const {
            PublicKeys k;
            k.n = n;
            k.rho = DevBuf(n * 32); k.tr = DevBuf(n * 64); k.t1 = DevBuf(n * (size_t)K * 1024);
            check(mldsa_get_public_key(Device::get().ctx(), SET, rho.template as<uint8_t>(), tr.template as<uint8_t>(),
                                       s1.template as<int32_t>(), s2.template as<int32_t>(), k.rho.template as<uint8_t>(),
                                       k.tr.template as<uint8_t>(), k.t1.template as<int32_t>(), n, nullptr), "mldsa_get_public_key");
            check(mldsa_stream_sync(nullptr), "sync");
            return k;
        }
        size_t n = 0;
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
                         sks.t0.template as<int32_t>(), sks.n, dk.as<uint32_t>(), m.bytes.as<uint8_t>(), m.offsets.as<uint64_t>(),
                         c.bytes.as<uint8_t>(), c.offsets.as<uint64_t>(), drnd.as<uint8_t>(), dsig.as<uint8_t>(),
                         dstat.as<int32_t>(), n, nullptr), "mldsa_sign");
        std::vector<int32_t> st(n);
        dstat.download(st.data(), n * 4);
        for (int32_t s : st)
            if (s == MLDSA_ERR_CTX_LEN) throw Error("ML-DSA.Sign: ctx too long");  // src/lib.rs:274
            else if (s != MLDSA_OK) throw Error("ML-DSA.Sign: key index out of range or malformed offsets");
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
                           pks.t1.template as<int32_t>(), pks.n, dk.as<uint32_t>(), m.bytes.as<uint8_t>(), m.offsets.as<uint64_t>(),
                           c.bytes.as<uint8_t>(), c.offsets.as<uint64_t>(), dsig.as<uint8_t>(), dok.as<uint8_t>(), n, nullptr),
              "mldsa_verify");
        std::vector<uint8_t> ok(n);
        dok.download(ok.data(), n);
        return std::vector<bool>(ok.begin(), ok.end());
    }

    // PublicKey::try_from_bytes + verify in one device call (mldsa_verify_pk): wire-format keys, one per op when key_idx is empty
    static std::vector<bool> verify_pk_many(const std::vector<PkBytes>& pk, const std::vector<uint32_t>& key_idx,
                                            const std::vector<std::vector<uint8_t>>& msgs, const std::vector<Signature>& sigs,
                                            const std::vector<std::vector<uint8_t>>& ctxs, int mode = MLDSA_MODE_PURE) {
        const size_t n = msgs.size();
        if (ctxs.size() != n || sigs.size() != n || (!key_idx.empty() && key_idx.size() != n)) throw Error("verify_pk_many: argument lengths differ");
        Packed m(msgs), c(ctxs);
        DevBuf dpk(pk.data(), pk.size() * PK), dk(key_idx.data(), key_idx.size() * 4), dsig(sigs.data(), n * SIG), dok(n);
        check(mldsa_verify_pk(Device::get().ctx(), SET, mode, dpk.as<uint8_t>(), pk.size(), key_idx.empty() ? nullptr : dk.as<uint32_t>(),
                              m.bytes.as<uint8_t>(), m.offsets.as<uint64_t>(), c.bytes.as<uint8_t>(), c.offsets.as<uint64_t>(), dsig.as<uint8_t>(),
                              dok.as<uint8_t>(), n, nullptr), "mldsa_verify_pk");
        std::vector<uint8_t> ok(n);
        dok.download(ok.data(), n);
        return std::vector<bool>(ok.begin(), ok.end());
    }

    // ---- the same three operations on host memory and wire-format keys (mldsa_*_host) --------------
    // Contiguous arrays in, contiguous arrays out; the library overlaps upload, kernels and download.
    struct HostBytes {  // concatenated byte strings + n + 1 offsets
        std::vector<uint8_t> flat;
        std::vector<uint64_t> off;
        explicit HostBytes(const std::vector<std::vector<uint8_t>>& items) : off(items.size() + 1, 0) {
            for (size_t i = 0; i < items.size(); i++) {
                flat.insert(flat.end(), items[i].begin(), items[i].end());
                off[i + 1] = flat.size();
            }
            if (flat.empty()) flat.push_back(0);
        }
    };
    static std::pair<std::vector<PkBytes>, std::vector<SkBytes>> keygen_host(const std::vector<std::array<uint8_t, 32>>& xi) {
        std::vector<PkBytes> pk(xi.size());
        std::vector<SkBytes> sk(xi.size());
        check(mldsa_keygen_host(Device::get().ctx(), SET, raw_bytes(xi), raw_bytes(pk), raw_bytes(sk), xi.size()),
              "mldsa_keygen_host");
        return {std::move(pk), std::move(sk)};
    }
    static std::vector<Signature> sign_host(const std::vector<SkBytes>& sk, const std::vector<uint32_t>& key_idx,
                                            const std::vector<std::vector<uint8_t>>& msgs, const std::vector<std::vector<uint8_t>>& ctxs,
                                            const std::vector<std::array<uint8_t, 32>>& rnd, int mode = MLDSA_MODE_PURE) {
        const size_t n = msgs.size();
        if (ctxs.size() != n || rnd.size() != n || key_idx.size() != n) throw Error("sign_host: argument lengths differ");
        HostBytes m(msgs), c(ctxs);
        std::vector<Signature> sig(n);
        std::vector<int32_t> st(n, 0);
        check(mldsa_sign_host(Device::get().ctx(), SET, mode, raw_bytes(sk), sk.size(), key_idx.data(), m.flat.data(), m.off.data(),
                              c.flat.data(), c.off.data(), raw_bytes(rnd), raw_bytes(sig), st.data(), n), "mldsa_sign_host");
        for (int32_t s : st)
            if (s == MLDSA_ERR_CTX_LEN) throw Error("ML-DSA.Sign: ctx too long");
            else if (s != MLDSA_OK) throw Error("ML-DSA.Sign: key index out of range");
        return sig;
    }
    static std::vector<bool> verify_host(const std::vector<PkBytes>& pk, const std::vector<uint32_t>& key_idx,
                                         const std::vector<std::vector<uint8_t>>& msgs, const std::vector<Signature>& sigs,
                                         const std::vector<std::vector<uint8_t>>& ctxs, int mode = MLDSA_MODE_PURE) {
        const size_t n = msgs.size();
        if (ctxs.size() != n || sigs.size() != n || key_idx.size() != n) throw Error("verify_host: argument lengths differ");
        HostBytes m(msgs), c(ctxs);
        std::vector<uint8_t> ok(n, 0);
        check(mldsa_verify_host(Device::get().ctx(), SET, mode, raw_bytes(pk), pk.size(), key_idx.data(), m.flat.data(), m.off.data(),
                                c.flat.data(), c.off.data(), raw_bytes(sigs), ok.data(), n), "mldsa_verify_host");
        return std::vector<bool>(ok.begin(), ok.end());
    }

    // ---- the same host-memory operations split over several GPUs (mldsa_group_*, include/mldsa_hip.h) ----------
    // Group({0, 1, ..., 7}): one context and one worker thread per device; contiguous ceil(B / N) slices; results identical
    // to the single-device calls above.
    class Group {
      public:
        explicit Group(const std::vector<int>& device_ids) {
            check(mldsa_group_create(device_ids.data(), (int)device_ids.size(), &g_), "mldsa_group_create");
        }
        ~Group() { mldsa_group_destroy(g_); }
        Group(const Group&) = delete;
        Group& operator=(const Group&) = delete;
        int size() const { return mldsa_group_size(g_); }
        mldsa_ctx* ctx(int i) const { return mldsa_group_ctx(g_, i); }
        std::pair<std::vector<PkBytes>, std::vector<SkBytes>> keygen_host(const std::vector<std::array<uint8_t, 32>>& xi) const {
            std::vector<PkBytes> pk(xi.size());
            std::vector<SkBytes> sk(xi.size());
            check(mldsa_keygen_host_group(g_, SET, raw_bytes(xi), raw_bytes(pk), raw_bytes(sk), xi.size()),
                  "mldsa_keygen_host_group");
            return {std::move(pk), std::move(sk)};
        }
        std::vector<Signature> sign_host(const std::vector<SkBytes>& sk, const std::vector<uint32_t>& key_idx,
                                         const std::vector<std::vector<uint8_t>>& msgs, const std::vector<std::vector<uint8_t>>& ctxs,
                                         const std::vector<std::array<uint8_t, 32>>& rnd, int mode = MLDSA_MODE_PURE) const {
            const size_t n = msgs.size();
            if (ctxs.size() != n || rnd.size() != n || key_idx.size() != n) throw Error("sign_host: argument lengths differ");
            HostBytes m(msgs), c(ctxs);
            std::vector<Signature> sig(n);
            std::vector<int32_t> st(n, 0);
            check(mldsa_sign_host_group(g_, SET, mode, raw_bytes(sk), sk.size(), key_idx.data(), m.flat.data(), m.off.data(),
                                        c.flat.data(), c.off.data(), raw_bytes(rnd), raw_bytes(sig), st.data(), n),
                  "mldsa_sign_host_group");
            for (int32_t s : st)
                if (s == MLDSA_ERR_CTX_LEN) throw Error("ML-DSA.Sign: ctx too long");
                else if (s != MLDSA_OK) throw Error("ML-DSA.Sign: key index out of range");
            return sig;
        }
        std::vector<bool> verify_host(const std::vector<PkBytes>& pk, const std::vector<uint32_t>& key_idx,
                                      const std::vector<std::vector<uint8_t>>& msgs, const std::vector<Signature>& sigs,
                                      const std::vector<std::vector<uint8_t>>& ctxs, int mode = MLDSA_MODE_PURE) const {
            const size_t n = msgs.size();
            if (ctxs.size() != n || sigs.size() != n || key_idx.size() != n) throw Error("verify_host: argument lengths differ");
            HostBytes m(msgs), c(ctxs);
            std::vector<uint8_t> ok(n, 0);
            check(mldsa_verify_host_group(g_, SET, mode, raw_bytes(pk), pk.size(), key_idx.data(), m.flat.data(), m.off.data(),
                                          c.flat.data(), c.off.data(), raw_bytes(sigs), ok.data(), n), "mldsa_verify_host_group");
            return std::vector<bool>(ok.begin(), ok.end());
        }
        // ---- device-resident slices (mldsa_*_group): slice i of the batch already lives on device i of the group -- keys
        // expanded there (PublicKeys / PrivateKeys created under Device::use(id)), inputs uploaded there.  One host thread
        // drives every device; wait = false returns once everything is enqueued (signing then has mldsa_sign_async
        // semantics) and sync() waits later.  ResidentSlice helpers fill the C structs from the key objects.
        static mldsa_verify_slice verify_slice(const PublicKeys& pks, const uint32_t* key_idx, const uint8_t* msgs, const uint64_t* msg_off,
                                               const uint8_t* ctxs, const uint64_t* ctx_off, const uint8_t* sigs, uint8_t* ok, size_t n_ops,
                                               void* stream = nullptr) {
            return mldsa_verify_slice{pks.rho.template as<uint8_t>(), pks.tr.template as<uint8_t>(), pks.t1.template as<int32_t>(), pks.n, key_idx,
                                      msgs, msg_off, ctxs, ctx_off, sigs, ok, n_ops, stream};
        }
        static mldsa_sign_slice sign_slice(const PrivateKeys& sks, const uint32_t* key_idx, const uint8_t* msgs, const uint64_t* msg_off,
                                           const uint8_t* ctxs, const uint64_t* ctx_off, const uint8_t* rnd, uint8_t* sigs, int32_t* status,
                                           size_t n_ops, void* stream = nullptr) {
            return mldsa_sign_slice{sks.rho.template as<uint8_t>(), sks.cap_k.template as<uint8_t>(), sks.tr.template as<uint8_t>(),
                                    sks.s1.template as<int32_t>(), sks.s2.template as<int32_t>(), sks.t0.template as<int32_t>(), sks.n, key_idx,
                                    msgs, msg_off, ctxs, ctx_off, rnd, sigs, status, n_ops, stream};
        }
        void verify_resident(const std::vector<mldsa_verify_slice>& slices, int mode = MLDSA_MODE_PURE, bool wait = true) const {
            if ((int)slices.size() != size()) throw Error("verify_resident: one slice per device of the group");
            check(mldsa_verify_group(g_, SET, mode, slices.data(), wait ? 1 : 0), "mldsa_verify_group");
        }
        void sign_resident(const std::vector<mldsa_sign_slice>& slices, int mode = MLDSA_MODE_PURE, bool wait = true) const {
            if ((int)slices.size() != size()) throw Error("sign_resident: one slice per device of the group");
            check(mldsa_sign_group(g_, SET, mode, slices.data(), wait ? 1 : 0), "mldsa_sign_group");
        }
        void keygen_resident(const std::vector<mldsa_keygen_slice>& slices, bool wait = true) const {
            if ((int)slices.size() != size()) throw Error("keygen_resident: one slice per device of the group");
            check(mldsa_keygen_group(g_, SET, slices.data(), wait ? 1 : 0), "mldsa_keygen_group");
        }
        void sync() const { check(mldsa_group_sync(g_), "mldsa_group_sync"); }
        // verdict bytes of every slice into every device's buffer (ordered after the group's last calls on the device)
        void allgather(const std::vector<uint8_t*>& bufs, size_t n_ops, int use_rccl = -1) const {
            if ((int)bufs.size() != size()) throw Error("allgather: one buffer per device of the group");
            check(mldsa_group_allgather(g_, bufs.data(), n_ops, use_rccl), "mldsa_group_allgather");
        }
        std::pair<size_t, size_t> shard(size_t n_ops, int part) const {
            size_t a = 0, c = 0;
            check(mldsa_group_shard(n_ops, size(), part, &a, &c), "mldsa_group_shard");
            return {a, c};
        }
      private:
        mldsa_group* g_ = nullptr;
    };

    // ---- single-operation calls from many host threads (mldsa_batcher_*) -----------------------------
    // The reference's one-call-per-operation surface -- pk.verify(message, sig, ctx), sk.try_sign_with_seed(rnd, message, ctx),
    // KG::keygen_from_seed(xi) -- for any number of threads at once: each call blocks its caller, the library coalesces the calls
    // that are in flight into batched ones, keeps every key's expanded form (and A_hat) in a device-resident table, and returns
    // each caller its own result.  Wire-format keys in, as a server holds them.
    class Batcher {
      public:
        explicit Batcher(size_t max_batch = 4096, unsigned max_wait_us = 0, size_t cache_keys = 0) {
            check(mldsa_batcher_create(Device::get().ctx(), SET, max_batch, max_wait_us, cache_keys, &b_), "mldsa_batcher_create");
        }
        ~Batcher() { mldsa_batcher_destroy(b_); }
        Batcher(const Batcher&) = delete;
        Batcher& operator=(const Batcher&) = delete;
        // Verifier::verify (src/lib.rs:364-380) after PublicKey::try_from_bytes
        bool verify(const PkBytes& pk, const std::vector<uint8_t>& message, const Signature& sig, const std::vector<uint8_t>& ctx,
                    int mode = MLDSA_MODE_PURE) const {
            uint8_t ok = 0;
            check(mldsa_batcher_verify(b_, mode, pk.data(), message.data(), message.size(), ctx.data(), ctx.size(), sig.data(), &ok),
                  "mldsa_batcher_verify");
            return ok != 0;
        }
        // Signer::try_sign_with_seed (src/lib.rs:268-296) after PrivateKey::try_from_bytes; throws for a ctx longer than 255 bytes
        Signature try_sign_with_seed(const SkBytes& sk, const std::array<uint8_t, 32>& rnd, const std::vector<uint8_t>& message,
                                     const std::vector<uint8_t>& ctx, int mode = MLDSA_MODE_PURE) const {
            Signature sig;
            const int rc = mldsa_batcher_sign(b_, mode, sk.data(), message.data(), message.size(), ctx.data(), ctx.size(), rnd.data(), sig.data());
            if (rc == MLDSA_ERR_CTX_LEN) throw Error("ML-DSA.Sign: ctx too long");
            check(rc, "mldsa_batcher_sign");
            return sig;
        }
        // KeyGen::keygen_from_seed (src/lib.rs:247-250)
        std::pair<PkBytes, SkBytes> keygen_from_seed(const std::array<uint8_t, 32>& xi) const {
            std::pair<PkBytes, SkBytes> out;
            check(mldsa_batcher_keygen(b_, xi.data(), out.first.data(), out.second.data()), "mldsa_batcher_keygen");
            return out;
        }
        // key lifetime: the table's copy of a key ends with the caller's (the reference's PrivateKey is ZeroizeOnDrop, src/types.rs:19)
        void forget_key(const SkBytes& sk) const { check(mldsa_batcher_forget_key(b_, sk.data(), sk.size()), "mldsa_batcher_forget_key"); }
        void forget_public_key(const PkBytes& pk) const { check(mldsa_batcher_forget_key(b_, pk.data(), pk.size()), "mldsa_batcher_forget_key"); }
        void flush_keys() const { check(mldsa_batcher_flush_keys(b_), "mldsa_batcher_flush_keys"); }
        void set_private_key_cache(bool on) const { check(mldsa_batcher_set_private_key_cache(b_, on ? 1 : 0), "mldsa_batcher_set_private_key_cache"); }
        mldsa_batcher_stats stats() const {
            mldsa_batcher_stats st;
            check(mldsa_batcher_get_stats(b_, &st), "mldsa_batcher_get_stats");
            return st;
        }
      private:
        mldsa_batcher* b_ = nullptr;
    };

    // ---- single-key objects with the reference's method names -------------------------------------
    class PublicKey {
      public:
        static PublicKey try_from_bytes(const PkBytes& pk) { return PublicKey(PublicKeys::try_from_bytes({pk})); }
        PkBytes into_bytes() const { return keys_->into_bytes()[0]; }
        // Verifier::verify (src/lib.rs:364-380)
        bool verify(const std::vector<uint8_t>& message, const Signature& sig, const std::vector<uint8_t>& ctx) const {
            return verify_many(*keys_, {0u}, {message}, {sig}, {ctx}, MLDSA_MODE_PURE)[0];
        }
        // _internal_verify (src/lib.rs:605-612)
        bool _internal_verify(const std::vector<uint8_t>& message, const Signature& sig, const std::vector<uint8_t>& ctx) const {
            return verify_many(*keys_, {0u}, {message}, {sig}, {ctx}, MLDSA_MODE_INTERNAL)[0];
        }
        // Verifier::hash_verify (src/traits.rs:361, src/lib.rs:391-411): HashML-DSA.Verify, PH(M) computed on the host
        bool hash_verify(const std::vector<uint8_t>& message, const Signature& sig, const std::vector<uint8_t>& ctx, Ph ph) const {
            return verify_many(*keys_, {0u}, {hash_message(message, ph)}, {sig}, {ctx}, MLDSA_MODE_PREHASH)[0];
        }
        explicit PublicKey(PublicKeys k) : keys_(std::make_shared<PublicKeys>(std::move(k))) {}
      private:
        std::shared_ptr<PublicKeys> keys_;
    };

    class PrivateKey {
      public:
        static PrivateKey try_from_bytes(const SkBytes& sk) { return PrivateKey(PrivateKeys::try_from_bytes({sk})); }
        SkBytes into_bytes() const { return keys_->into_bytes()[0]; }
        // Signer::get_public_key (src/lib.rs:345-349)
        auto get_public_key() const;
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
        // Signer::try_sign (src/traits.rs:156-158) / try_hash_sign (247-251): rnd from OsRng
        Signature try_sign(const std::vector<uint8_t>& message, const std::vector<uint8_t>& ctx) const {
            OsRng rng;
            return try_sign_with_rng(rng, message, ctx);
        }
        Signature try_hash_sign(const std::vector<uint8_t>& message, const std::vector<uint8_t>& ctx, Ph ph) const {
            OsRng rng;
            return try_hash_sign_with_rng(rng, message, ctx, ph);
        }
        // Signer::try_hash_sign_with_seed / try_hash_sign_with_rng (src/traits.rs:265-284, src/lib.rs:310-342): HashML-DSA.Sign
        Signature try_hash_sign_with_seed(const std::array<uint8_t, 32>& rnd, const std::vector<uint8_t>& message,
                                          const std::vector<uint8_t>& ctx, Ph ph) const {
            return sign_many(*keys_, {0u}, {hash_message(message, ph)}, {ctx}, {rnd}, MLDSA_MODE_PREHASH)[0];
        }
        template <class Rng>
        Signature try_hash_sign_with_rng(Rng& rng, const std::vector<uint8_t>& message, const std::vector<uint8_t>& ctx, Ph ph) const {
            if (ctx.size() > 255) throw Error("HashML-DSA.Sign: ctx too long");  // before the rng is touched, lib.rs:316
            std::array<uint8_t, 32> rnd{};
            if (!rng.try_fill_bytes(rnd.data(), 32)) throw Error("HashML-DSA.Sign: random number generator failed");
            return try_hash_sign_with_seed(rnd, message, ctx, ph);
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
        // KeyGen::try_keygen (src/traits.rs:44-46): xi <- OsRng
        static std::pair<PublicKey, PrivateKey> try_keygen() {
            OsRng rng;
            return try_keygen_with_rng(rng);
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

template <int SET, int K_, int L_, size_t PK, size_t SK, size_t SIG>
auto ParamSet<SET, K_, L_, PK, SK, SIG>::PrivateKey::get_public_key() const {
    return typename ParamSet<SET, K_, L_, PK, SK, SIG>::PublicKey(keys_->get_public_key());
}

using ml_dsa_44 = ParamSet<MLDSA_44, 4, 4, 1312, 2560, 2420>;
using ml_dsa_65 = ParamSet<MLDSA_65, 6, 5, 1952, 4032, 3309>;
using ml_dsa_87 = ParamSet<MLDSA_87, 8, 7, 2592, 4896, 4627>;

}  // namespace fips204_hip
