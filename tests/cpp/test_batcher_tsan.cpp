// The real csrc/batcher.cpp over stand-ins for the device (stub_cabi.cpp, stub_hip.cpp), built with -fsanitize=thread: 24 threads of
// single-operation calls -- three parameter-set-independent paths (verify, sign, keygen), three modes, 40 keys over a 16-slot key
// table, batches of at most 8, one and three lanes -- every result compared with what the stand-in's batched entry points give for the
// same arguments.  ThreadSanitizer reports any unsynchronised access in the batcher's queues, counters and wake-ups.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "../../include/mldsa_hip.h"

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } } while (0)

int main() {
    const int set = 65;
    mldsa_params p;
    REQUIRE(mldsa_get_params(set, &p) == 0);
    mldsa_ctx *ctx = nullptr;
    REQUIRE(mldsa_ctx_create(0, &ctx) == 0);
    const size_t NK = 40, NR = 200;
    std::vector<uint8_t> xi(NK * 32), pk(NK * p.pk_len), sk(NK * p.sk_len);
    std::mt19937_64 rng(204);
    for (auto &b : xi) b = (uint8_t)rng();
    REQUIRE(mldsa_keygen_host(ctx, set, xi.data(), pk.data(), sk.data(), NK) == 0);
    struct Reqs { uint32_t key; int mode; std::vector<uint8_t> msg, ctx, rnd, sig; };
    std::vector<Reqs> reqs(NR);
    for (size_t i = 0; i < NR; i++) {
        Reqs &r = reqs[i];
        r.key = (uint32_t)(rng() % NK);
        r.mode = (int)(i % 3);
        r.msg.resize(rng() % 300);
        r.ctx.resize(r.mode == MLDSA_MODE_INTERNAL ? 0 : rng() % 40);
        r.rnd.resize(32);
        for (auto &b : r.msg) b = (uint8_t)rng();
        for (auto &b : r.ctx) b = (uint8_t)rng();
        for (auto &b : r.rnd) b = (uint8_t)rng();
        r.sig.resize((size_t)p.sig_len);
        const uint64_t mo[2] = {0, r.msg.size()}, co[2] = {0, r.ctx.size()};
        const uint8_t one = 0;
        REQUIRE(mldsa_sign_host(ctx, set, r.mode, sk.data() + r.key * (size_t)p.sk_len, 1, nullptr, r.msg.empty() ? &one : r.msg.data(), mo,
                                r.ctx.empty() ? &one : r.ctx.data(), co, r.rnd.data(), r.sig.data(), nullptr, 1) == 0);
    }
    for (int lanes : {1, 3}) {
        mldsa_batcher *b = nullptr;
        if (lanes == 1) REQUIRE(mldsa_batcher_create(ctx, set, 8, 0, 16, &b) == 0);
        else { const int ids[3] = {0, 1, 0}; REQUIRE(mldsa_batcher_create_on(ids, 3, set, 8, 50, 16, &b) == 0); }
        REQUIRE(mldsa_batcher_lanes(b) == lanes);
        std::atomic<uint64_t> calls{0};
        std::vector<std::thread> th;
        for (int t = 0; t < 24; t++)
            th.emplace_back([&, t] {
                std::mt19937_64 r(1000 + (uint64_t)t);
                std::vector<uint8_t> sig((size_t)p.sig_len), pk1((size_t)p.pk_len), sk1((size_t)p.sk_len);
                for (int it = 0; it < 150; it++) {
                    const Reqs &q = reqs[r() % NR];
                    const unsigned what = (unsigned)(r() % 10);
                    uint8_t ok = 9;
                    if (what < 4) {
                        REQUIRE(mldsa_batcher_sign(b, q.mode, sk.data() + q.key * (size_t)p.sk_len, q.msg.data(), q.msg.size(), q.ctx.data(), q.ctx.size(),
                                                   q.rnd.data(), sig.data()) == 0);
                        REQUIRE(sig == q.sig);
                    } else if (what < 7) {
                        REQUIRE(mldsa_batcher_verify(b, q.mode, pk.data() + q.key * (size_t)p.pk_len, q.msg.data(), q.msg.size(), q.ctx.data(), q.ctx.size(),
                                                     q.sig.data(), &ok) == 0);
                        REQUIRE(ok == 1);
                    } else if (what < 9) {
                        const uint32_t other = (q.key + 1) % (uint32_t)NK;
                        REQUIRE(mldsa_batcher_verify(b, q.mode, pk.data() + other * (size_t)p.pk_len, q.msg.data(), q.msg.size(), q.ctx.data(), q.ctx.size(),
                                                     q.sig.data(), &ok) == 0);
                        REQUIRE(ok == 0);
                    } else {
                        const size_t k = r() % NK;
                        REQUIRE(mldsa_batcher_keygen(b, xi.data() + 32 * k, pk1.data(), sk1.data()) == 0);
                        REQUIRE(!std::memcmp(pk1.data(), pk.data() + k * (size_t)p.pk_len, pk1.size()) && !std::memcmp(sk1.data(), sk.data() + k * (size_t)p.sk_len, sk1.size()));
                    }
                    calls++;
                }
            });
        for (auto &x : th) x.join();
        mldsa_batcher_stats st;
        REQUIRE(mldsa_batcher_get_stats(b, &st) == 0);
        REQUIRE(st.requests == calls.load() && st.batches <= st.requests && st.largest_batch <= 8 && st.keys_expanded > 80);
        // a ctx of 256 bytes never reaches a batch (lib.rs:274, 368)
        std::vector<uint8_t> long_ctx(256, 1), sig((size_t)p.sig_len);
        uint8_t ok = 1;
        REQUIRE(mldsa_batcher_verify(b, MLDSA_MODE_PURE, pk.data(), nullptr, 0, long_ctx.data(), 256, reqs[0].sig.data(), &ok) == 0 && ok == 0);
        REQUIRE(mldsa_batcher_sign(b, MLDSA_MODE_PURE, sk.data(), nullptr, 0, long_ctx.data(), 256, reqs[0].rnd.data(), sig.data()) == MLDSA_ERR_CTX_LEN);
        std::printf("lanes %d: %llu calls in %llu batches, largest %llu, %llu keys expanded, %llu found\n", lanes, (unsigned long long)st.requests,
                    (unsigned long long)st.batches, (unsigned long long)st.largest_batch, (unsigned long long)st.keys_expanded, (unsigned long long)st.key_hits);
        mldsa_batcher_destroy(b);
    }
    mldsa_ctx_destroy(ctx);
    std::printf("OK\n");
    return 0;
}
