// The C++ host mirror (fips204_amd/host/fips204_hip.hpp, prehash.hpp) driven over the stub C ABI (stub_cabi.cpp) under
// AddressSanitizer + UndefinedBehaviorSanitizer: argument packing (ragged messages, ctxs, key indices, offsets), the
// device-buffer RAII, the group calls' slice arithmetic, the error mapping -- and the mirror's own SHA-256 / SHA-512 /
// SHAKE128 against vectors tests/test_sanitizers_cpu.py computes with hashlib and passes on stdin.
// Mirrors the reference's debug-assertions + overflow-checks fuzz profile (fuzz/Cargo.toml:31-35) for the C++ side.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "../../fips204_amd/host/fips204_hip.hpp"

using namespace fips204_hip;
#define ASSERT(cond) do { if (!(cond)) { std::fprintf(stderr, "ASSERT FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)

static std::vector<uint8_t> from_hex(const std::string& h) {
    std::vector<uint8_t> v(h.size() / 2);
    for (size_t i = 0; i < v.size(); i++) v[i] = (uint8_t)std::stoul(h.substr(2 * i, 2), nullptr, 16);
    return v;
}
static std::vector<uint8_t> pattern(size_t n, unsigned seed) {
    std::vector<uint8_t> v(n);
    for (size_t i = 0; i < n; i++) v[i] = (uint8_t)(seed * 31u + i * 7u + (i >> 8));
    return v;
}

template <class F>
static bool throws0(F&& f) { try { f(); } catch (const fips204_hip::Error&) { return true; } return false; }

template <class P>
static void plumbing() {
    const size_t nk = 5, n = 23;
    std::vector<std::array<uint8_t, 32>> xi(nk);
    for (size_t i = 0; i < nk; i++) xi[i].fill((uint8_t)(i + 1));
    auto keys = P::keygen_many(xi);
    ASSERT(keys.first.size() == nk && keys.second.size() == nk);
    auto keys_h = P::keygen_host(xi);
    ASSERT(keys_h.first == keys.first && keys_h.second == keys.second);
    // SerDes round trips through the expanded, field-by-field device form
    auto pks = P::PublicKeys::try_from_bytes(keys.first);
    auto sks = P::PrivateKeys::try_from_bytes(keys.second);
    ASSERT(pks.into_bytes() == keys.first);
    ASSERT(sks.into_bytes() == keys.second);
    auto derived = sks.get_public_key();
    ASSERT(derived.n == nk);
    // ragged messages (empty, 1 byte, a block boundary, long), ctxs 0 ... 255 bytes, scattered key indices
    std::vector<std::vector<uint8_t>> msgs(n), ctxs(n);
    std::vector<uint32_t> kidx(n);
    std::vector<std::array<uint8_t, 32>> rnd(n);
    const size_t mlens[] = {0, 1, 135, 136, 137, 1000, 4097};
    for (size_t i = 0; i < n; i++) {
        msgs[i] = pattern(mlens[i % 7], (unsigned)i);
        ctxs[i] = pattern(i == 7 ? 255 : (i * 11) % 64, (unsigned)(i + 100));
        kidx[i] = (uint32_t)((i * 3) % nk);
        rnd[i].fill((uint8_t)i);
    }
    for (int mode : {MLDSA_MODE_PURE, MLDSA_MODE_INTERNAL, MLDSA_MODE_PREHASH}) {
        auto sig = P::sign_many(sks, kidx, msgs, ctxs, rnd, mode);
        auto ok = P::verify_many(pks, kidx, msgs, sig, ctxs, mode);
        for (size_t i = 0; i < n; i++) ASSERT(ok[i]);
        auto sig_h = P::sign_host(keys.second, kidx, msgs, ctxs, rnd, mode);
        ASSERT(sig_h == sig);
        auto ok_h = P::verify_host(keys.first, kidx, msgs, sig, ctxs, mode);
        for (size_t i = 0; i < n; i++) ASSERT(ok_h[i]);
        auto ok_pk = P::verify_pk_many(keys.first, kidx, msgs, sig, ctxs, mode);   // try_from_bytes + verify in one call
        for (size_t i = 0; i < n; i++) ASSERT(ok_pk[i]);
        // one flipped bit, a swapped message, a wrong key, a different ctx, a different mode: each op on its own
        auto bad = sig;
        bad[3][P::SIG_LEN - 1] ^= 1;
        auto msgs2 = msgs; std::swap(msgs2[5], msgs2[6]);
        auto kidx2 = kidx; kidx2[9] = (kidx[9] + 1) % nk;
        auto ctxs2 = ctxs; ctxs2[11].push_back(0);
        auto v1 = P::verify_many(pks, kidx, msgs, bad, ctxs, mode);
        auto v2 = P::verify_many(pks, kidx, msgs2, sig, ctxs, mode);
        auto v3 = P::verify_many(pks, kidx2, msgs, sig, ctxs, mode);
        auto v4 = P::verify_many(pks, kidx, msgs, sig, ctxs2, mode);
        auto v5 = P::verify_many(pks, kidx, msgs, sig, ctxs, (mode + 1) % 3);
        for (size_t i = 0; i < n; i++) {
            ASSERT(v1[i] == (i != 3));
            ASSERT(v2[i] == (i != 5 && i != 6));
            ASSERT(v3[i] == (i != 9));
            ASSERT(v4[i] == (i != 11));
            ASSERT(!v5[i]);
        }
        // groups: same bytes whatever the number of slices, ragged (23 % 4 != 0) and more slices than ops
        for (std::vector<int> devs : {std::vector<int>{0, 1}, std::vector<int>{0, 0, 1, 1}, std::vector<int>(1, 0)}) {
            typename P::Group g(devs);
            ASSERT(g.size() == (int)devs.size() && g.ctx(0) != nullptr && g.ctx((int)devs.size()) == nullptr);
            ASSERT(g.sign_host(keys.second, kidx, msgs, ctxs, rnd, mode) == sig);
            auto okg = g.verify_host(keys.first, kidx, msgs, bad, ctxs, mode);
            for (size_t i = 0; i < n; i++) ASSERT(okg[i] == (i != 3));
            auto kg = g.keygen_host(xi);
            ASSERT(kg.first == keys.first && kg.second == keys.second);
        }
    }
    {   // two ops over a group of four: two slices are empty
        typename P::Group g({0, 0, 1, 1});
        std::vector<std::vector<uint8_t>> m2(msgs.begin(), msgs.begin() + 2), c2(ctxs.begin(), ctxs.begin() + 2);
        std::vector<uint32_t> k2(kidx.begin(), kidx.begin() + 2);
        std::vector<std::array<uint8_t, 32>> r2(rnd.begin(), rnd.begin() + 2);
        auto s2 = g.sign_host(keys.second, k2, m2, c2, r2);
        auto o2 = g.verify_host(keys.first, k2, m2, s2, c2);
        ASSERT(o2.size() == 2 && o2[0] && o2[1]);
    }
    {   // device-resident slices through the group: 23 ops over 3 contexts (ragged) and 2 ops over 4 (two empty slices)
        for (std::vector<int> devs : {std::vector<int>{0, 1, 1}, std::vector<int>{0, 0, 1, 1}}) {
            typename P::Group g(devs);
            const size_t nn = devs.size() == 3 ? n : 2;
            const int N = g.size();
            struct Staged { Packed m, c; DevBuf k, r, sig, st, ok; };
            std::vector<std::unique_ptr<Staged>> st;
            std::vector<mldsa_sign_slice> ss;
            std::vector<mldsa_verify_slice> vs;
            for (int i = 0; i < N; i++) {
                auto sh = g.shard(nn, i);
                const size_t a = sh.first, c = sh.second;
                std::vector<std::vector<uint8_t>> mi(msgs.begin() + a, msgs.begin() + a + c), ci(ctxs.begin() + a, ctxs.begin() + a + c);
                st.emplace_back(new Staged{Packed(mi), Packed(ci), DevBuf(kidx.data() + a, c * 4), DevBuf(raw_bytes(rnd) + 32 * a, c * 32),
                                           DevBuf(c * P::SIG_LEN), DevBuf(c * 4), DevBuf(c)});
                Staged& t = *st.back();
                ss.push_back(P::Group::sign_slice(sks, t.k.template as<uint32_t>(), t.m.bytes.template as<uint8_t>(), t.m.offsets.template as<uint64_t>(),
                                                  t.c.bytes.template as<uint8_t>(), t.c.offsets.template as<uint64_t>(), t.r.template as<uint8_t>(),
                                                  t.sig.template as<uint8_t>(), t.st.template as<int32_t>(), c));
                vs.push_back(P::Group::verify_slice(pks, t.k.template as<uint32_t>(), t.m.bytes.template as<uint8_t>(), t.m.offsets.template as<uint64_t>(),
                                                    t.c.bytes.template as<uint8_t>(), t.c.offsets.template as<uint64_t>(), t.sig.template as<uint8_t>(),
                                                    t.ok.template as<uint8_t>(), c));
            }
            g.sign_resident(ss, MLDSA_MODE_PURE, false);
            g.sync();
            g.verify_resident(vs);
            auto want = P::sign_many(sks, kidx, msgs, ctxs, rnd);
            for (int i = 0; i < N; i++) {
                auto sh = g.shard(nn, i);
                std::vector<typename P::Signature> got(sh.second);
                st[(size_t)i]->sig.download(got.data(), sh.second * P::SIG_LEN);
                std::vector<uint8_t> ok(sh.second);
                st[(size_t)i]->ok.download(ok.data(), sh.second);
                for (size_t j = 0; j < sh.second; j++) ASSERT(got[j] == want[sh.first + j] && ok[j] == 1);
            }
            ASSERT(throws0([&] { g.verify_resident(std::vector<mldsa_verify_slice>(1)); }) == (N != 1));  // one slice per device
        }
    }
    // error mapping: an over-long ctx and an out-of-range key index are per-op statuses that the mirror turns into Errors
    auto throws = [](auto&& f) { try { f(); } catch (const Error&) { return true; } return false; };
    auto long_ctx = ctxs; long_ctx[2] = pattern(256, 1);
    ASSERT(throws([&] { (void)P::sign_many(sks, kidx, msgs, long_ctx, rnd); }));
    ASSERT(throws([&] { (void)P::sign_host(keys.second, kidx, msgs, long_ctx, rnd); }));
    auto far_key = kidx; far_key[4] = (uint32_t)nk;
    ASSERT(throws([&] { (void)P::sign_many(sks, far_key, msgs, ctxs, rnd); }));
    ASSERT(!P::verify_many(pks, far_key, msgs, P::sign_many(sks, kidx, msgs, ctxs, rnd), ctxs)[4]);   // verify: false, never an error
    ASSERT(!P::verify_many(pks, kidx, msgs, P::sign_many(sks, kidx, msgs, ctxs, rnd), long_ctx)[2]);
    ASSERT(throws([&] { (void)P::sign_many(sks, kidx, msgs, ctxs, std::vector<std::array<uint8_t, 32>>(n - 1)); }));  // lengths differ
    ASSERT(throws([&] { typename P::Group g(std::vector<int>{0, 7}); }));                                              // no such device
    // empty batches
    ASSERT(P::sign_many(sks, {}, {}, {}, {}).empty());
    ASSERT(P::verify_host(keys.first, {}, {}, {}, {}).empty());
    // single-key objects with the reference's method names
    auto kp = P::KG::keygen_from_seed(xi[0]);
    const std::vector<uint8_t> message = pattern(77, 9), ctx = pattern(13, 3);
    auto sig1 = kp.second.try_sign_with_seed(rnd[1], message, ctx);
    ASSERT(kp.first.verify(message, sig1, ctx) && !kp.first.verify(message, sig1, {}));
    ASSERT(P::PublicKey::try_from_bytes(kp.first.into_bytes()).verify(message, sig1, ctx));
    ASSERT(P::PrivateKey::try_from_bytes(kp.second.into_bytes()).try_sign_with_seed(rnd[1], message, ctx) == sig1);
    for (Ph ph : {Ph::SHA256, Ph::SHA512, Ph::SHAKE128}) {
        auto hs = kp.second.try_hash_sign_with_seed(rnd[2], message, ctx, ph);
        ASSERT(kp.first.hash_verify(message, hs, ctx, ph));
        ASSERT(!kp.first.verify(message, hs, ctx));
    }
    ASSERT(kp.first._internal_verify(message, kp.second._internal_sign(message, ctx, rnd[3]), ctx));
}

int main() {
    // stdin: "<ph> <hex message> <hex OID || digest>" lines from hashlib
    std::string ph, mhex, want;
    size_t n_vec = 0;
    while (std::cin >> ph >> mhex >> want) {
        const Ph p = ph == "SHA256" ? Ph::SHA256 : ph == "SHA512" ? Ph::SHA512 : Ph::SHAKE128;
        const std::vector<uint8_t> msg = from_hex(mhex == "-" ? "" : mhex);
        if (hash_message(msg, p) != from_hex(want)) { std::fprintf(stderr, "hash_message mismatch: %s len %zu\n", ph.c_str(), msg.size()); return 1; }
        n_vec++;
    }
    ASSERT(n_vec > 0);
    plumbing<ml_dsa_44>();
    plumbing<ml_dsa_65>();
    plumbing<ml_dsa_87>();
    std::printf("OK %zu hash vectors, 3 parameter sets\n", n_vec);
    return 0;
}
