// Opening a round of the signer's rejection loop on the device: shared by k_make_slots (kernels_sign.hip), k_compact_small (the next round
// of a small call, same file) and the finisher of k_sign_prologue_small (round 0 of a small call, kernels_small.hip).
#pragma once
#include "ctx.h"

namespace mldsa {

// What opens a round (kernels_sign.hip, "Speculative rounds"), for `nthreads` threads numbered `tid`; thread 0 writes the control block.  m = unfinished ops entering
// the round, spec = the rule's candidates per op for that m, prev_gen = candidates per op the PREVIOUS round generated (gen_par of the
// other parity).  fresh: the control block was zero a moment ago (the small prologue clears it in the same launch): the statistics are
// written, not added to.
__device__ __forceinline__ void make_slots_body(RoundCtl* __restrict__ ctl, int parity, uint32_t m, uint32_t spec, uint32_t prev_gen, uint32_t ns_cap,
                                                const uint32_t* __restrict__ act, const uint16_t* __restrict__ kappa, int l,
                                                uint32_t* __restrict__ slot_op, uint16_t* __restrict__ slot_kappa,
                                                const uint32_t* __restrict__ key_idx, uint32_t* __restrict__ gen_op,
                                                uint16_t* __restrict__ gen_kappa, uint32_t* __restrict__ gen_key, int may_use_pre,
                                                int may_gen2, const uint32_t* __restrict__ ypos, uint32_t* __restrict__ slot_y, uint32_t tid,
                                                uint32_t nthreads, bool fresh) {
    // ns_cap = the slots the workspace was carved for (plan_sign: the rule's maximum over every m, >= the batch): never exceeded,
    // whatever rule and count arrive here
    if (m && (unsigned long long)m * spec > ns_cap) spec = ns_cap / m ? ns_cap / m : 1u;
    const uint32_t ns = m * spec;
    const bool use_pre = may_use_pre && spec == 1u && prev_gen == 2u;
    const uint32_t gen = use_pre ? 0u : (may_gen2 && spec == 1u) ? 2u : 1u;
    const uint32_t ns_gen = gen == 2u ? 2u * m : gen == 1u ? ns : 0u;
    if (tid == 0) {
        ctl->cnt[parity ^ 1] = 0;
        ctl->m = m;
        ctl->m_par[parity] = m;
        ctl->spec = spec;
        ctl->gen_par[parity] = gen;
        ctl->use_pre = use_pre ? 1u : 0u;
        ctl->ns = ns;
        ctl->ns_gen = ns_gen;
        // candidates generated (statistics: a second candidate counts whether or not it is ever tested)
        ctl->slots_total = (fresh ? 0ull : ctl->slots_total) + ns_gen;
        ctl->ops_total = (fresh ? 0ull : ctl->ops_total) + (gen ? m : 0u);
        ctl->rounds = (fresh ? 0u : ctl->rounds) + (m ? 1u : 0u);
    }
    const uint32_t top = ns > ns_gen ? ns : ns_gen;
    for (uint32_t sidx = tid; sidx < top; sidx += nthreads) {
        if (sidx < ns) {  // the candidates this round tests
            const uint32_t op = act[sidx / spec];
            slot_op[sidx] = op;
            slot_kappa[sidx] = (uint16_t)(kappa[op] + (sidx % spec) * (uint32_t)l);
            slot_y[sidx] = use_pre ? 2u * ypos[sidx] + 1u : gen == 2u ? 2u * sidx : sidx;
        }
        if (sidx < ns_gen) {  // the rows this round generates
            const uint32_t per = gen == 2u ? 2u : spec;
            const uint32_t op = act[sidx / per];
            gen_op[sidx] = op;
            gen_kappa[sidx] = (uint16_t)(kappa[op] + (sidx % per) * (uint32_t)l);
            if (gen_key) gen_key[sidx] = key_idx ? key_idx[op] : op;  // row of a per-key A_hat table
        }
    }
}

}  // namespace mldsa
