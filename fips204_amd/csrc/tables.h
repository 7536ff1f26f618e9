// Host-side generation of the per-lane twiddle tables consumed by ntt_wave.h.
#pragma once
#include <stdint.h>
#include <vector>

namespace mldsa {

typedef int32_t HostTwiddle;

// ZETA_TABLE_MONT of the reference (src/helpers.rs:171-184): table[brv8(i)] = zeta^i * 2^32 mod q
void gen_zeta_table_mont(int32_t out[256]);

// [FWD_TW][64] and [INV_TW][64] tables; throws std::logic_error if the simulated register
// layout does not end in the layout ntt_wave.h documents.
std::vector<HostTwiddle> gen_fwd_lane_twiddles();
std::vector<HostTwiddle> gen_inv_lane_twiddles();

}  // namespace mldsa
