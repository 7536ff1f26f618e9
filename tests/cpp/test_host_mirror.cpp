// GPU test of the C++ host mirror (fips204_amd/host/fips204_hip.hpp).  Reads like the reference's
// own tests: src/lib.rs:497-552 (smoke_test per parameter set), tests/integration.rs:79-119
// (test_44_no_verif), tests/messages.rs:10-21 (fixed vector; xi / rnd and the expected hex are
// passed on the command line by tests/test_gpu_cpp_host.py, which also checks them against the
// golden fixture).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

#include "../../fips204_amd/host/fips204_hip.hpp"

using namespace fips204_hip;

#define ASSERT(cond) do { if (!(cond)) { std::fprintf(stderr, "ASSERT FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)

// deterministic replay RNG, like TestRng of tests/nist_vectors/mod.rs:23-52
struct CountingRng {
    uint8_t next = 1;
    bool try_fill_bytes(uint8_t* out, size_t n) {
        for (size_t i = 0; i < n; i++) out[i] = (uint8_t)(next * 131u + (uint8_t)i * 7u);
        next++;
        return true;
    }
};
struct FailingRng {
    bool try_fill_bytes(uint8_t*, size_t) { return false; }
};

static std::vector<uint8_t> from_hex(const std::string& h) {
    std::vector<uint8_t> v(h.size() / 2);
    for (size_t i = 0; i < v.size(); i++) v[i] = (uint8_t)std::stoul(h.substr(2 * i, 2), nullptr, 16);
    return v;
}
static std::string to_hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) { s.push_back(d[p[i] >> 4]); s.push_back(d[p[i] & 15]); }
    return s;
}

template <class P>
static void smoke_test(int expect_pk0) {  // src/lib.rs:497-552
    CountingRng rng;
    const std::vector<uint8_t> message1 = {0, 1, 2, 3, 4, 5, 6, 7}, message2 = {7, 7, 7, 7, 7, 7, 7, 7};
    for (int i = 0; i < 4; i++) {
        auto [pk, sk] = P::KG::try_keygen_with_rng(rng);
        auto sig = sk.try_sign_with_rng(rng, message1, {});
        ASSERT(pk.verify(message1, sig, {}));
        ASSERT(!pk.verify(message2, sig, {}));
        ASSERT(P::PublicKey::try_from_bytes(pk.into_bytes()).verify(message1, sig, {}));  // SerDes round trip
    }
    auto [pk, sk] = P::KG::keygen_from_seed([] { std::array<uint8_t, 32> a; a.fill(0x11); return a; }());
    std::array<uint8_t, 32> seed12; seed12.fill(12);
    auto sig = sk.try_sign_with_seed(seed12, message1, {});
    ASSERT(pk.verify(message1, sig, {}));
    ASSERT(!pk.verify(message1, sig, std::vector<uint8_t>(257, 0)));              // lib.rs:527
    bool threw = false;
    try { (void)sk.try_sign_with_seed(seed12, message1, std::vector<uint8_t>(257, 0)); } catch (const Error&) { threw = true; }
    ASSERT(threw);                                                                 // lib.rs:528
    // HashML-DSA with every pre-hash function (lib.rs:510-541)
    for (Ph ph : {Ph::SHA256, Ph::SHA512, Ph::SHAKE128}) {
        auto hsig = sk.try_hash_sign_with_rng(rng, message1, {}, ph);
        ASSERT(pk.hash_verify(message1, hsig, {}, ph));
        ASSERT(!pk.hash_verify(message2, hsig, {}, ph));
        ASSERT(!pk.verify(message1, hsig, {}));                                    // domain separation: not a pure signature
        ASSERT(!pk.hash_verify(message1, hsig, {}, ph == Ph::SHA256 ? Ph::SHAKE128 : Ph::SHA256));  // same length, other OID
    }
    std::array<uint8_t, 32> seed34; seed34.fill(34);
    auto hsig = sk.try_hash_sign_with_seed(seed34, message1, {}, Ph::SHA256);     // lib.rs:539-540
    ASSERT(pk.hash_verify(message1, hsig, {}, Ph::SHA256));
    threw = false;
    try { (void)sk.try_hash_sign_with_seed(seed34, message1, std::vector<uint8_t>(256, 0), Ph::SHA512); } catch (const Error&) { threw = true; }
    ASSERT(threw);                                                                 // lib.rs:316
    ASSERT(!pk.hash_verify(message1, hsig, std::vector<uint8_t>(256, 0), Ph::SHA256));  // lib.rs:395-397
    ASSERT(pk.into_bytes()[0] == expect_pk0);                                      // lib.rs:543-545
    // Signer::get_public_key (lib.rs:345-349): the public key derived from the private key is the one keygen gave
    ASSERT(sk.get_public_key().into_bytes() == pk.into_bytes());
    ASSERT(sk.get_public_key().verify(message1, sig, {}));
    // SerDes round trip of the private key through its expanded form (lib.rs:421-465)
    ASSERT(P::PrivateKey::try_from_bytes(sk.into_bytes()).into_bytes() == sk.into_bytes());
    {   // the OsRng entry points (traits.rs:44-46, 156-158, 247-251): hedged, so two signatures of one message differ
        auto [pk2, sk2] = P::KG::try_keygen();
        auto s1 = sk2.try_sign(message1, {}), s2 = sk2.try_sign(message1, {});
        ASSERT(pk2.verify(message1, s1, {}) && pk2.verify(message1, s2, {}) && s1 != s2);
        ASSERT(pk2.hash_verify(message1, sk2.try_hash_sign(message1, {}, Ph::SHA512), {}, Ph::SHA512));
        ASSERT(pk2.into_bytes() != pk.into_bytes());
    }
    FailingRng bad;
    threw = false;
    try { (void)P::KG::try_keygen_with_rng(bad); } catch (const Error&) { threw = true; }
    ASSERT(threw);                                                                 // ml_dsa.rs:41
}

static void test_44_no_verif() {  // tests/integration.rs:79-119
    using P = ml_dsa_44;
    CountingRng rng;
    const std::vector<uint8_t> msg = {0, 1, 2, 3, 4, 5, 6, 7}, ctx = {0};
    auto [pk, sk] = P::KG::try_keygen_with_rng(rng);
    auto sig = sk.try_sign_with_rng(rng, msg, ctx);
    ASSERT(pk.verify(msg, sig, ctx));
    for (int i = 0; i < 8; i++) {
        auto bad = msg; bad[i] ^= 0x08;
        ASSERT(!pk.verify(bad, sig, ctx));
    }
    for (int i = 0; i < 8; i++) {
        auto skb = sk.into_bytes(); skb[70 + i * 10] ^= 0x08;
        auto sig2 = P::PrivateKey::try_from_bytes(skb).try_sign_with_rng(rng, msg, ctx);
        ASSERT(!pk.verify(msg, sig2, ctx));
    }
    for (int i = 0; i < 8; i++) {
        auto pkb = pk.into_bytes(); pkb[i * 10] ^= 0x08;
        ASSERT(!P::PublicKey::try_from_bytes(pkb).verify(msg, sig, ctx));
    }
    for (int i = 0; i < 8; i++) {
        auto s2 = sig; s2[i * 10] ^= 0x08;
        ASSERT(!pk.verify(msg, s2, ctx));
    }
}

int main(int argc, char** argv) {
    smoke_test<ml_dsa_44>(197);
    smoke_test<ml_dsa_65>(177);
    smoke_test<ml_dsa_87>(16);
    test_44_no_verif();
    if (argc >= 3) {  // tests/messages.rs:10-21 with xi, rnd from ChaCha8Rng::seed_from_u64(123)
        using P = ml_dsa_44;
        std::array<uint8_t, 32> xi{}, rnd{};
        auto a = from_hex(argv[1]), b = from_hex(argv[2]);
        ASSERT(a.size() == 32 && b.size() == 32);
        std::memcpy(xi.data(), a.data(), 32);
        std::memcpy(rnd.data(), b.data(), 32);
        auto [pk, sk] = P::KG::keygen_from_seed(xi);
        auto sig = sk.try_sign_with_seed(rnd, {'a', 's', 'd', 'f'}, {});
        ASSERT(pk.verify({'a', 's', 'd', 'f'}, sig, {}));
        auto skb = sk.into_bytes(); auto pkb = pk.into_bytes();
        std::printf("sk %s\nsig %s\npk %s\n", to_hex(skb.data(), skb.size()).c_str(), to_hex(sig.data(), sig.size()).c_str(),
                    to_hex(pkb.data(), pkb.size()).c_str());
        // HashML-DSA under the same key and seed, one line per pre-hash function: the driver compares them with the oracle
        const std::vector<uint8_t> ctx = {'c', 't', 'x'};
        const char* names[] = {"SHA256", "SHA512", "SHAKE128"};
        int pi = 0;
        for (Ph ph : {Ph::SHA256, Ph::SHA512, Ph::SHAKE128}) {
            auto hs = sk.try_hash_sign_with_seed(rnd, {'a', 's', 'd', 'f'}, ctx, ph);
            ASSERT(pk.hash_verify({'a', 's', 'd', 'f'}, hs, ctx, ph));
            std::printf("hsig_%s %s\n", names[pi++], to_hex(hs.data(), hs.size()).c_str());
        }
    }
    // a small batch through the *_many calls: 3 keys, 12 ops, one corrupted signature
    {
        using P = ml_dsa_65;
        std::vector<std::array<uint8_t, 32>> xi(3), rnd(12);
        for (int i = 0; i < 3; i++) xi[i].fill((uint8_t)(40 + i));
        for (int i = 0; i < 12; i++) rnd[i].fill((uint8_t)i);
        auto ks = P::keygen_many(xi);
        auto pks = P::PublicKeys::try_from_bytes(ks.first);
        auto sks = P::PrivateKeys::try_from_bytes(ks.second);
        std::vector<uint32_t> kidx(12);
        std::vector<std::vector<uint8_t>> msgs(12), ctxs(12);
        for (int i = 0; i < 12; i++) { kidx[i] = (uint32_t)(i % 3); msgs[i].assign((size_t)(i * 37), (uint8_t)i); ctxs[i].assign((size_t)(i % 4), 9); }
        auto sigs = P::sign_many(sks, kidx, msgs, ctxs, rnd);
        sigs[5][100] ^= 1;
        auto ok = P::verify_many(pks, kidx, msgs, sigs, ctxs);
        for (int i = 0; i < 12; i++) ASSERT(ok[i] == (i != 5));
    }
    // the same batch through the host-memory entry points (wire-format keys, staging inside the library)
    {
        using P = ml_dsa_87;
        const size_t n = 40;
        std::vector<std::array<uint8_t, 32>> xi(4), rnd(n);
        for (int i = 0; i < 4; i++) xi[i].fill((uint8_t)(90 + i));
        for (size_t i = 0; i < n; i++) rnd[i].fill((uint8_t)(3 * i));
        auto ks = P::keygen_host(xi);
        auto ks_dev = P::keygen_many(xi);
        ASSERT(ks.first == ks_dev.first && ks.second == ks_dev.second);
        std::vector<uint32_t> kidx(n);
        std::vector<std::vector<uint8_t>> msgs(n), ctxs(n);
        for (size_t i = 0; i < n; i++) { kidx[i] = (uint32_t)(i % 4); msgs[i].assign(i * 11, (uint8_t)i); ctxs[i].assign(i % 3, 5); }
        auto sigs = P::sign_host(ks.second, kidx, msgs, ctxs, rnd);
        auto sigs_dev = P::sign_many(P::PrivateKeys::try_from_bytes(ks.second), kidx, msgs, ctxs, rnd);
        ASSERT(sigs == sigs_dev);
        sigs[7][2000] ^= 4;
        auto ok = P::verify_host(ks.first, kidx, msgs, sigs, ctxs);
        for (size_t i = 0; i < n; i++) ASSERT(ok[i] == (i != 7));
    }
    // device-resident slices through a group of two contexts (both on GPU 0): one host thread, mldsa_sign_group without
    // waiting + mldsa_group_sync, mldsa_verify_group, verdicts gathered with mldsa_group_allgather right behind it
    {
        using P = ml_dsa_44;
        const size_t n = 37;
        std::vector<std::array<uint8_t, 32>> xi(3), rnd(n);
        for (int i = 0; i < 3; i++) xi[i].fill((uint8_t)(7 + i));
        for (size_t i = 0; i < n; i++) rnd[i].fill((uint8_t)(5 * i + 1));
        auto ks = P::keygen_many(xi);
        auto pks = P::PublicKeys::try_from_bytes(ks.first);
        auto sks = P::PrivateKeys::try_from_bytes(ks.second);
        std::vector<uint32_t> kidx(n);
        std::vector<std::vector<uint8_t>> msgs(n), ctxs(n);
        for (size_t i = 0; i < n; i++) { kidx[i] = (uint32_t)(i % 3); msgs[i].assign(i * 5, (uint8_t)(i + 1)); ctxs[i].assign(i % 6, 2); }
        auto want = P::sign_many(sks, kidx, msgs, ctxs, rnd);
        P::Group g({0, 0});
        struct Staged { Packed m, c; DevBuf k, r, sig, st; };
        std::vector<std::unique_ptr<Staged>> st;
        std::vector<mldsa_sign_slice> ss;
        std::vector<mldsa_verify_slice> vs;
        const size_t per = (n + 1) / 2;
        std::vector<DevBuf> okb;
        for (int i = 0; i < 2; i++) okb.emplace_back(2 * per);
        for (int i = 0; i < 2; i++) {
            auto sh = g.shard(n, i);
            const size_t a = sh.first, c = sh.second;
            std::vector<std::vector<uint8_t>> mi(msgs.begin() + a, msgs.begin() + a + c), ci(ctxs.begin() + a, ctxs.begin() + a + c);
            st.emplace_back(new Staged{Packed(mi), Packed(ci), DevBuf(kidx.data() + a, c * 4), DevBuf(raw_bytes(rnd) + 32 * a, c * 32),
                                       DevBuf(c * P::SIG_LEN), DevBuf(c * 4)});
            Staged& t = *st.back();
            ss.push_back(P::Group::sign_slice(sks, t.k.as<uint32_t>(), t.m.bytes.as<uint8_t>(), t.m.offsets.as<uint64_t>(), t.c.bytes.as<uint8_t>(),
                                              t.c.offsets.as<uint64_t>(), t.r.as<uint8_t>(), t.sig.as<uint8_t>(), t.st.as<int32_t>(), c));
            vs.push_back(P::Group::verify_slice(pks, t.k.as<uint32_t>(), t.m.bytes.as<uint8_t>(), t.m.offsets.as<uint64_t>(), t.c.bytes.as<uint8_t>(),
                                                t.c.offsets.as<uint64_t>(), t.sig.as<uint8_t>(), okb[(size_t)i].as<uint8_t>() + a, c));
        }
        g.sign_resident(ss, MLDSA_MODE_PURE, false);
        g.sync();
        for (int i = 0; i < 2; i++) {
            auto sh = g.shard(n, i);
            std::vector<P::Signature> got(sh.second);
            st[(size_t)i]->sig.download(got.data(), sh.second * P::SIG_LEN);
            for (size_t j = 0; j < sh.second; j++) ASSERT(got[j] == want[sh.first + j]);
        }
        g.verify_resident(vs, MLDSA_MODE_PURE, false);
        g.allgather({okb[0].as<uint8_t>(), okb[1].as<uint8_t>()}, n, 0);
        for (int i = 0; i < 2; i++) {
            std::vector<uint8_t> ok(n);
            okb[(size_t)i].download(ok.data(), n);
            for (size_t j = 0; j < n; j++) ASSERT(ok[j] == 1);
        }
        ASSERT(mldsa_abi_version() == MLDSA_ABI_VERSION);
    }
    // single-operation calls from 24 host threads through a Batcher: every caller gets the signature / verdict the batched
    // calls give for its own arguments (try_sign_with_seed is deterministic given rnd)
    {
        using P = ml_dsa_65;
        const size_t n = 96;
        std::vector<std::array<uint8_t, 32>> xi(4), rnd(n);
        for (int i = 0; i < 4; i++) xi[i].fill((uint8_t)(40 + i));
        for (size_t i = 0; i < n; i++) rnd[i].fill((uint8_t)(3 * i + 2));
        auto ks = P::keygen_many(xi);
        auto sks = P::PrivateKeys::try_from_bytes(ks.second);
        std::vector<uint32_t> kidx(n);
        std::vector<std::vector<uint8_t>> msgs(n), ctxs(n);
        for (size_t i = 0; i < n; i++) { kidx[i] = (uint32_t)(i % 4); msgs[i].assign(1 + i * 3, (uint8_t)i); ctxs[i].assign(i % 5, 9); }
        auto want = P::sign_many(sks, kidx, msgs, ctxs, rnd);
        P::Batcher b(32);
        std::vector<P::Signature> got(n);
        std::vector<int> good(n, 0), bad(n, 1);
        std::vector<std::thread> th;
        for (int t = 0; t < 24; t++)
            th.emplace_back([&, t] {
                for (size_t i = (size_t)t; i < n; i += 24) {
                    got[i] = b.try_sign_with_seed(ks.second[kidx[i]], rnd[i], msgs[i], ctxs[i]);
                    good[i] = b.verify(ks.first[kidx[i]], msgs[i], got[i], ctxs[i]);
                    bad[i] = b.verify(ks.first[(kidx[i] + 1) % 4], msgs[i], got[i], ctxs[i]);
                }
            });
        for (auto& x : th) x.join();
        for (size_t i = 0; i < n; i++) ASSERT(got[i] == want[i] && good[i] == 1 && bad[i] == 0);
        auto kp = b.keygen_from_seed(xi[2]);
        ASSERT(kp.first == ks.first[2] && kp.second == ks.second[2]);
        const mldsa_batcher_stats st = b.stats();
        ASSERT(st.requests == 3 * n + 1 && st.batches < st.requests && st.keys_expanded == 8);
        bool threw = false;
        try { b.try_sign_with_seed(ks.second[0], rnd[0], msgs[0], std::vector<uint8_t>(256, 1)); } catch (const Error&) { threw = true; }
        ASSERT(threw);
    }
    std::printf("OK\n");
    return 0;
}
