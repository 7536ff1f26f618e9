// Host-side scatter of 65 536 signature rows (3 309 B) from a contiguous staging buffer into op order: what a DMA-engine export of
// mldsa_sign_host would need on the host.  g++ -O2 -pthread tools/ubench_host_scatter.cpp -o /tmp/scat; /tmp/scat <threads>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
int main(int argc, char** argv) {
    const size_t n = 65536, sl = 3309;
    int T = argc > 1 ? atoi(argv[1]) : 1;
    std::vector<uint8_t> src(n * sl, 1), dst(n * sl, 0);
    std::vector<uint32_t> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::mt19937 g(1);
    std::shuffle(idx.begin(), idx.end(), g);
    for (int rep = 0; rep < 3; rep++) {
        auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] { for (size_t i = t; i < n; i += T) memcpy(&dst[(size_t)idx[i] * sl], &src[i * sl], sl); });
        for (auto& x : th) x.join();
        double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("threads %d: %.2f ms, %.1f GB/s\n", T, dt * 1e3, n * sl / dt / 1e9);
    }
}
