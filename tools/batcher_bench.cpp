// (a container's CPU quota is not visible here: with a quota of 16 CPUs, as on the pool's boxes, points above ~64 threads measure the
// quota's throttling -- p99 of tens of milliseconds -- not the batcher)
// Many host threads making the reference's ONE-operation calls (pk.verify(msg, sig, ctx), sk.try_sign_with_seed(..), src/lib.rs:268-296,
// 364-380) through mldsa_batcher_* -- calls per second, batch sizes the library formed, latency of a call as its caller sees it --
// next to the same calls made one at a time with n_ops = 1 (mldsa_verify_host / mldsa_sign_host), which is what a shim without a
// batcher would do.
//
//   g++ -O2 -std=c++17 -I include tools/batcher_bench.cpp -o /tmp/batcher_bench -L fips204_amd/csrc -lmldsa_hip \
//       -Wl,-rpath,$PWD/fips204_amd/csrc -Wl,-rpath,/opt/rocm/lib -lpthread
//   /tmp/batcher_bench [set = 65] [seconds per point = 2] [max_wait_us = 0] [threads = 1,8,64,256] [lanes = 1]   -> one JSON object on stdout
// lanes > 1: mldsa_batcher_create_on({0, 0, ...}): that many dispatchers (contexts) on GPU 0
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mldsa_hip.h"

#define CHECK(e) do { int rc_ = (e); if (rc_ != MLDSA_OK) { fprintf(stderr, "%s: %d %s\n", #e, rc_, mldsa_last_error()); exit(1); } } while (0)

using clk = std::chrono::steady_clock;
static double secs(clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); }

struct Point { int threads; double calls_per_s, mean_batch, p50_us, p99_us, batches_per_s; uint64_t largest; };

int main(int argc, char **argv) {
    const int set = argc > 1 ? atoi(argv[1]) : 65;
    const double dur = argc > 2 ? atof(argv[2]) : 2.0;
    const unsigned max_wait = argc > 3 ? (unsigned)atoi(argv[3]) : 0;
    std::vector<int> thread_counts;
    for (const char *q = argc > 4 ? argv[4] : "1,8,64,256"; *q;) {
        thread_counts.push_back(atoi(q));
        while (*q && *q != ',') q++;
        if (*q == ',') q++;
    }
    const int lanes = argc > 5 ? atoi(argv[5]) : 1;
    mldsa_params p;
    CHECK(mldsa_get_params(set, &p));
    mldsa_ctx *ctx;
    CHECK(mldsa_ctx_create(0, &ctx));
    // a pool of NK keys and NM signed 32-byte messages (32-byte messages, empty ctx)
    const size_t NK = 64, NM = 4096;
    std::vector<uint8_t> xi(NK * 32), pk(NK * p.pk_len), sk(NK * p.sk_len), msgs(NM * 32), rnd(NM * 32), sigs(NM * (size_t)p.sig_len), ok(NM);
    std::vector<uint32_t> kidx(NM);
    std::vector<uint64_t> off(NM + 1);
    srand(204);
    for (auto &b : xi) b = (uint8_t)rand();
    for (auto &b : msgs) b = (uint8_t)rand();
    for (auto &b : rnd) b = (uint8_t)rand();
    for (size_t i = 0; i <= NM; i++) off[i] = 32 * i;
    for (size_t i = 0; i < NM; i++) kidx[i] = (uint32_t)(i % NK);
    CHECK(mldsa_keygen_host(ctx, set, xi.data(), pk.data(), sk.data(), NK));
    CHECK(mldsa_sign_host(ctx, set, MLDSA_MODE_PURE, sk.data(), NK, kidx.data(), msgs.data(), off.data(), nullptr, nullptr, rnd.data(), sigs.data(), nullptr, NM));
    CHECK(mldsa_verify_host(ctx, set, MLDSA_MODE_PURE, pk.data(), NK, kidx.data(), msgs.data(), off.data(), nullptr, nullptr, sigs.data(), ok.data(), NM));
    for (size_t i = 0; i < NM; i++) if (!ok[i]) { fprintf(stderr, "setup: signature %zu does not verify\n", i); return 1; }

    // ---- one call at a time, n_ops = 1 (no batcher)
    double direct_us[2];
    for (int op = 0; op < 2; op++) {
        const int reps = 300;
        std::vector<uint8_t> s1((size_t)p.sig_len);
        uint8_t ok1;
        const uint64_t o2[2] = {0, 32};
        const auto t0 = clk::now();
        for (int r = 0; r < reps; r++) {
            const size_t i = (size_t)r % NM;
            if (op == 0) CHECK(mldsa_verify_host(ctx, set, MLDSA_MODE_PURE, pk.data() + kidx[i] * (size_t)p.pk_len, 1, nullptr, msgs.data() + 32 * i, o2, nullptr, nullptr, sigs.data() + i * (size_t)p.sig_len, &ok1, 1));
            else CHECK(mldsa_sign_host(ctx, set, MLDSA_MODE_PURE, sk.data() + kidx[i] * (size_t)p.sk_len, 1, nullptr, msgs.data() + 32 * i, o2, nullptr, nullptr, rnd.data() + 32 * i, s1.data(), nullptr, 1));
        }
        direct_us[op] = secs(t0, clk::now()) / reps * 1e6;
    }

    std::string json = "{\"tool\": \"tools/batcher_bench.cpp\", \"set\": " + std::to_string(set) + ", \"seconds_per_point\": " + std::to_string(dur) +
                       ", \"max_wait_us\": " + std::to_string(max_wait) + ", \"lanes\": " + std::to_string(lanes) + ", \"keys\": " + std::to_string(NK) + ", \"host_cpus_online\": " + std::to_string(std::thread::hardware_concurrency()) +
                       ", \"one_call_at_a_time_n_ops_1\": {\"verify_us\": " + std::to_string(direct_us[0]) + ", \"sign_us\": " + std::to_string(direct_us[1]) +
                       ", \"verify_calls_per_s\": " + std::to_string(1e6 / direct_us[0]) + ", \"sign_calls_per_s\": " + std::to_string(1e6 / direct_us[1]) + "}";
    const char *names[2] = {"verify", "sign"};
    for (int op = 0; op < 2; op++) {
        json += std::string(", \"") + names[op] + "\": [";
        bool first = true;
        for (int T : thread_counts) {
            mldsa_batcher *b;
            if (lanes <= 1) CHECK(mldsa_batcher_create(ctx, set, 8192, max_wait, 0, &b));
            else { std::vector<int> ids((size_t)lanes, 0); CHECK(mldsa_batcher_create_on(ids.data(), lanes, set, 8192, max_wait, 0, &b)); }
            std::atomic<bool> go{false}, stop{false};
            std::atomic<uint64_t> bad{0};
            std::vector<std::vector<float>> lat((size_t)T);
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++)
                th.emplace_back([&, t] {
                    std::vector<uint8_t> s1((size_t)p.sig_len);
                    uint8_t ok1 = 0;
                    size_t i = ((size_t)t * 2654435761u) % NM;
                    while (!go.load()) std::this_thread::yield();
                    while (!stop.load()) {
                        const auto a = clk::now();
                        int rc;
                        if (op == 0) rc = mldsa_batcher_verify(b, MLDSA_MODE_PURE, pk.data() + kidx[i] * (size_t)p.pk_len, msgs.data() + 32 * i, 32, nullptr, 0, sigs.data() + i * (size_t)p.sig_len, &ok1);
                        else rc = mldsa_batcher_sign(b, MLDSA_MODE_PURE, sk.data() + kidx[i] * (size_t)p.sk_len, msgs.data() + 32 * i, 32, nullptr, 0, rnd.data() + 32 * i, s1.data());
                        lat[(size_t)t].push_back((float)(secs(a, clk::now()) * 1e6));
                        if (rc != MLDSA_OK || (op == 0 && !ok1) || (op == 1 && memcmp(s1.data(), sigs.data() + i * (size_t)p.sig_len, (size_t)p.sig_len))) {
                            if (bad++ == 0) fprintf(stderr, "first wrong result: rc = %d (%s)\n", rc, rc != MLDSA_OK ? mldsa_last_error() : "result differs");
                        }
                        i = (i + 977) % NM;
                    }
                });
            // warm-up, then the timed window
            go = true;
            std::this_thread::sleep_for(std::chrono::milliseconds(300));
            mldsa_batcher_stats s0, s1;
            CHECK(mldsa_batcher_get_stats(b, &s0));
            const auto t0 = clk::now();
            std::this_thread::sleep_for(std::chrono::duration<double>(dur));
            CHECK(mldsa_batcher_get_stats(b, &s1));
            const double el = secs(t0, clk::now());
            stop = true;
            for (auto &x : th) x.join();
            mldsa_batcher_destroy(b);
            if (bad.load()) { fprintf(stderr, "%s, %d threads: %llu calls returned a wrong result\n", names[op], T, (unsigned long long)bad.load()); return 1; }
            std::vector<float> all;
            for (auto &v : lat) all.insert(all.end(), v.begin(), v.end());
            std::sort(all.begin(), all.end());
            const uint64_t reqs = s1.requests - s0.requests, batches = s1.batches - s0.batches;
            char buf[512];
            snprintf(buf, sizeof buf, "%s{\"threads\": %d, \"calls_per_s\": %.0f, \"batches_per_s\": %.0f, \"mean_batch\": %.1f, \"largest_batch\": %llu, \"p50_us\": %.0f, \"p99_us\": %.0f}",
                     first ? "" : ", ", T, reqs / el, batches / el, batches ? (double)reqs / batches : 0.0, (unsigned long long)s1.largest_batch,
                     all.empty() ? 0.0 : all[all.size() / 2], all.empty() ? 0.0 : all[(size_t)(all.size() * 0.99)]);
            json += buf;
            first = false;
        }
        json += "]";
    }
    json += "}";
    puts(json.c_str());
    mldsa_ctx_destroy(ctx);
    return 0;
}
