// Per-lane twiddle tables for the wave-level NTT (see ntt_wave.h).  The tables are derived
// by replaying, on index numbers, the exact register-exchange sequence the kernels run,
// so the element <-> (lane, register) map is never written down by hand.
#include "tables.h"

#include <stdexcept>
#include <utility>

namespace mldsa {

static constexpr int64_t Q64 = 8380417;

void gen_zeta_table_mont(int32_t out[256]) {
    int64_t x = 1;
    for (unsigned i = 0; i < 256; i++) {
        unsigned r = 0;
        for (int b = 0; b < 8; b++) r |= ((i >> b) & 1u) << (7 - b);
        out[r] = (int32_t)((x << 32) % Q64);
        x = (x * 1753) % Q64;
    }
}

namespace {

struct Layout {
    int idx[64][4];
    // upper lane's register a <-> lower lane's register b, lanes l and l | m
    void swap_pair(int m, int a, int b) {
        for (int l = 0; l < 64; l++)
            if (!(l & m)) std::swap(idx[l | m][a], idx[l][b]);
    }
    void xchg_hi(int m) { swap_pair(m, 0, 2); swap_pair(m, 1, 3); }
    void xchg_lo(int m) { swap_pair(m, 0, 1); swap_pair(m, 2, 3); }
};

HostTwiddle make_tw(int64_t z) { return (HostTwiddle)z; }

}  // namespace

std::vector<HostTwiddle> gen_fwd_lane_twiddles() {
    int32_t zeta[256];
    gen_zeta_table_mont(zeta);
    Layout L;
    for (int l = 0; l < 64; l++)
        for (int k = 0; k < 4; k++) L.idx[l][k] = 64 * k + l;
    std::vector<HostTwiddle> tab;
    // levels s = 7, 6 are in-lane with lane-uniform twiddles (constants in ntt_wave.h)
    for (int s = 5; s >= 0; s--) {
        const bool hi = (s & 1) != 0;
        if (hi) L.xchg_hi(1 << s); else L.xchg_lo(1 << s);
        const int pa[2] = {0, hi ? 1 : 2};
        const int pb[2] = {hi ? 2 : 1, 3};
        for (int bf = 0; bf < 2; bf++)
            for (int l = 0; l < 64; l++) {
                int a = L.idx[l][pa[bf]], b = L.idx[l][pb[bf]];
                if (b != a + (1 << s) || (a & (1 << s))) throw std::logic_error("fwd NTT layout: bad pair");
                int m = (128 >> s) + (a >> (s + 1));  // ntt.rs:39-42
                tab.push_back(make_tw(zeta[m]));
            }
    }
    for (int l = 0; l < 64; l++)
        for (int k = 0; k < 4; k++)
            if (L.idx[l][k] != 4 * l + k) throw std::logic_error("fwd NTT layout: bad final layout");
    return tab;
}

std::vector<HostTwiddle> gen_inv_lane_twiddles() {
    int32_t zeta[256];
    gen_zeta_table_mont(zeta);
    Layout L;
    for (int l = 0; l < 64; l++)
        for (int k = 0; k < 4; k++) L.idx[l][k] = 4 * l + k;
    std::vector<HostTwiddle> tab;
    for (int s = 0; s <= 6; s++) {
        const bool hi = (s & 1) != 0;
        if (s >= 2) { if (hi) L.xchg_hi(1 << (s - 2)); else L.xchg_lo(1 << (s - 2)); }
        const int pa[2] = {0, hi ? 1 : 2};
        const int pb[2] = {hi ? 2 : 1, 3};
        for (int bf = 0; bf < 2; bf++)
            for (int l = 0; l < 64; l++) {
                int a = L.idx[l][pa[bf]], b = L.idx[l][pb[bf]];
                if (b != a + (1 << s) || (a & (1 << s))) throw std::logic_error("inv NTT layout: bad pair");
                int nb = 128 >> s;
                int m = 2 * nb - 1 - (a >> (s + 1));  // ntt.rs:111-117: m counts down from 255
                tab.push_back(make_tw(-(int64_t)zeta[m]));
            }
    }
    L.xchg_hi(32);  // s = 7, uniform twiddle -zeta[1]
    for (int l = 0; l < 64; l++)
        for (int k = 0; k < 4; k++)
            if (L.idx[l][k] != 64 * k + l) throw std::logic_error("inv NTT layout: bad final layout");
    return tab;
}

}  // namespace mldsa
