//! Raw FFI binding of `libmldsa_hip.so` (include/mldsa_hip.h): the MI355X batched ML-DSA hot path behind a C ABI.
//!
//! GENERATED from the header by tools/gen_rust_sys.py -- do not edit; tests/test_rust_binding_cpu.py pins it to the header
//! (names, arity, argument classes, struct field order and widths, constants).  This crate is the ONE place with `unsafe`
//! declarations, so that `#![deny(unsafe_code)]` (fips204 src/lib.rs:2) keeps holding for the reference crate itself; the seams it
//! stands behind are the crate-private imports of src/ml_dsa.rs:3-11 and the constants of src/lib.rs:118-124.
#![no_std]
#![allow(non_camel_case_types)]
use core::ffi::{c_char, c_int, c_long, c_uint, c_ulonglong, c_void};

#[repr(C)]
pub struct mldsa_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mldsa_group {
    _private: [u8; 0],
}
#[repr(C)]
pub struct mldsa_batcher {
    _private: [u8; 0],
}

pub const MLDSA_OK: c_int = 0;
pub const MLDSA_ERR_PARAM: c_int = -1;
pub const MLDSA_ERR_CTX_LEN: c_int = -2;
pub const MLDSA_ERR_DEVICE: c_int = -3;
pub const MLDSA_ERR_NOMEM: c_int = -4;
pub const MLDSA_ERR_AGAIN: c_int = -5;
pub const MLDSA_44: c_int = 44;
pub const MLDSA_65: c_int = 65;
pub const MLDSA_87: c_int = 87;
pub const MLDSA_MODE_PURE: c_int = 0;
pub const MLDSA_MODE_INTERNAL: c_int = 1;
pub const MLDSA_MODE_PREHASH: c_int = 2;
pub const MLDSA_ABI_VERSION: c_int = 6;
pub const MLDSA_OP_KEYGEN: c_int = 1;
pub const MLDSA_OP_SIGN: c_int = 2;
pub const MLDSA_OP_VERIFY: c_int = 3;
pub const MLDSA_OPT_GRAPHS: c_int = 1;
pub const MLDSA_OPT_SPEC_TARGET: c_int = 2;
pub const MLDSA_OPT_SPEC_MAX: c_int = 3;
pub const MLDSA_OPT_VA_BLOCKS_PER_CU: c_int = 4;
pub const MLDSA_OPT_GRAPH_CACHE: c_int = 5;
pub const MLDSA_OPT_SIGN_ROUNDS: c_int = 6;
pub const MLDSA_OPT_SIGN_LANES: c_int = 7;
pub const MLDSA_OPT_SIGN_CT0_EXACT: c_int = 8;
pub const MLDSA_OPT_SIGN_ASYNC_EXP: c_int = 9;
pub const MLDSA_OPT_SIGN_LOOKAHEAD: c_int = 10;
pub const MLDSA_OPT_WORKSPACE_CAP_MB: c_int = 11;
pub const MLDSA_OPT_COOP_HASH: c_int = 12;
pub const MLDSA_OPT_SMALL_FUSED: c_int = 13;
pub const MLDSA_REDUCE_PARTIAL: c_int = 0;
pub const MLDSA_REDUCE_FULL: c_int = 1;
pub const MLDSA_REDUCE_CENTER: c_int = 2;
pub const MLDSA_ROUND_POWER2ROUND: c_int = 0;
pub const MLDSA_ROUND_DECOMPOSE: c_int = 1;
pub const MLDSA_ROUND_HIGH_BITS: c_int = 2;
pub const MLDSA_ROUND_LOW_BITS: c_int = 3;
pub const MLDSA_ROUND_MAKE_HINT: c_int = 4;
pub const MLDSA_ROUND_USE_HINT: c_int = 5;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mldsa_params {
    pub set: c_int,
    pub k: c_int,
    pub l: c_int,
    pub eta: c_int,
    pub tau: c_int,
    pub lambda: c_int,
    pub gamma1: c_int,
    pub gamma2: c_int,
    pub omega: c_int,
    pub beta: c_int,
    pub ctilde_len: c_int,
    pub pk_len: c_int,
    pub sk_len: c_int,
    pub sig_len: c_int,
    pub w1_len: c_int,
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mldsa_stats {
    pub graphs_captured: c_ulonglong,
    pub graph_replays: c_ulonglong,
    pub direct_calls: c_ulonglong,
    pub workspace_growths: c_ulonglong,
    pub sign_extra_rounds: c_ulonglong,
    pub workspace_shrinks: c_ulonglong,
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mldsa_verify_slice {
    pub rho: *const u8,
    pub tr: *const u8,
    pub t1_d2_hat_mont: *const i32,
    pub n_keys: usize,
    pub key_idx: *const u32,
    pub msgs: *const u8,
    pub msg_off: *const u64,
    pub ctxs: *const u8,
    pub ctx_off: *const u64,
    pub sigs: *const u8,
    pub ok: *mut u8,
    pub n_ops: usize,
    pub stream: *mut c_void,
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mldsa_sign_slice {
    pub rho: *const u8,
    pub cap_k: *const u8,
    pub tr: *const u8,
    pub s_1_hat_mont: *const i32,
    pub s_2_hat_mont: *const i32,
    pub t_0_hat_mont: *const i32,
    pub n_keys: usize,
    pub key_idx: *const u32,
    pub msgs: *const u8,
    pub msg_off: *const u64,
    pub ctxs: *const u8,
    pub ctx_off: *const u64,
    pub rnd: *const u8,
    pub sigs: *mut u8,
    pub status: *mut i32,
    pub n_ops: usize,
    pub stream: *mut c_void,
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mldsa_keygen_slice {
    pub xi: *const u8,
    pub pk: *mut u8,
    pub sk: *mut u8,
    pub n_keys: usize,
    pub stream: *mut c_void,
}
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct mldsa_batcher_stats {
    pub batches: u64,
    pub requests: u64,
    pub largest_batch: u64,
    pub keys_expanded: u64,
    pub key_hits: u64,
}

#[link(name = "mldsa_hip")]
extern "C" {
    pub fn mldsa_abi_version() -> c_int;
    pub fn mldsa_ctx_create(device_id: c_int, out: *mut *mut mldsa_ctx) -> c_int;
    pub fn mldsa_ctx_destroy(ctx: *mut mldsa_ctx);
    pub fn mldsa_last_error() -> *const c_char;
    pub fn mldsa_get_params(set: c_int, out: *mut mldsa_params) -> c_int;
    pub fn mldsa_device_count() -> c_int;
    pub fn mldsa_ctx_device(ctx: *const mldsa_ctx) -> c_int;
    pub fn mldsa_reserve(ctx: *mut mldsa_ctx, set: c_int, op: c_int, n_ops: usize) -> c_int;
    pub fn mldsa_ctx_set_workspace(ctx: *mut mldsa_ctx, dev_buf: *mut c_void, bytes: usize) -> c_int;
    pub fn mldsa_set_option(ctx: *mut mldsa_ctx, option: c_int, value: c_long) -> c_int;
    pub fn mldsa_get_option(ctx: *const mldsa_ctx, option: c_int) -> c_long;
    pub fn mldsa_get_stats(ctx: *mut mldsa_ctx, out: *mut mldsa_stats) -> c_int;
    pub fn mldsa_get_stats_sized(ctx: *mut mldsa_ctx, out: *mut c_void, out_bytes: usize) -> c_int;
    pub fn mldsa_profile_enable(ctx: *mut mldsa_ctx, on: c_int) -> c_int;
    pub fn mldsa_profile_report(ctx: *mut mldsa_ctx, buf: *mut c_char, buf_len: usize) -> c_int;
    pub fn mldsa_debug_secret_residue(ctx: *mut mldsa_ctx, scanned_bytes: *mut usize, nonzero_bytes: *mut usize) -> c_int;
    pub fn mldsa_debug_count_nonzero(dev_ptr: *const c_void, bytes: usize, nonzero: *mut usize) -> c_int;
    pub fn mldsa_malloc(dev_ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn mldsa_ctx_malloc(ctx: *mut mldsa_ctx, dev_ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn mldsa_free(dev_ptr: *mut c_void) -> c_int;
    pub fn mldsa_host_alloc(host_ptr: *mut *mut c_void, bytes: usize) -> c_int;
    pub fn mldsa_host_free(host_ptr: *mut c_void) -> c_int;
    pub fn mldsa_memcpy_h2d(dst_dev: *mut c_void, src_host: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_memcpy_d2h(dst_host: *mut c_void, src_dev: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_memset(dst_dev: *mut c_void, value: c_int, bytes: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_stream_sync(stream: *mut c_void) -> c_int;
    pub fn mldsa_ntt(ctx: *mut mldsa_ctx, w: *const i32, w_hat: *mut i32, n_polys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_inv_ntt(ctx: *mut mldsa_ctx, w_hat: *const i32, w: *mut i32, n_polys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_to_mont(ctx: *mut mldsa_ctx, in_: *const i32, out: *mut i32, n_polys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_reduce(ctx: *mut mldsa_ctx, kind: c_int, in_: *const i32, out: *mut i32, n_polys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_rounding(ctx: *mut mldsa_ctx, set: c_int, op: c_int, a: *const i32, b: *const i32, out1: *mut i32, out2: *mut i32, n_polys: usize,
        stream: *mut c_void) -> c_int;
    pub fn mldsa_xof(ctx: *mut mldsa_ctx, bits: c_int, data: *const u8, off: *const u64, out: *mut u8, out_len: usize, bad: *mut u8, n_ops: usize,
        stream: *mut c_void) -> c_int;
    pub fn mldsa_bit_pack(ctx: *mut mldsa_ctx, w: *const i32, a: c_int, b: c_int, out: *mut u8, n_polys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_bit_unpack(ctx: *mut mldsa_ctx, v: *const u8, a: c_int, b: c_int, w: *mut i32, ok: *mut u8, n_polys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_hint_bit_pack(ctx: *mut mldsa_ctx, set: c_int, h: *const i32, y: *mut u8, ok: *mut u8, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_hint_bit_unpack(ctx: *mut mldsa_ctx, set: c_int, y: *const u8, h: *mut i32, ok: *mut u8, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_sig_encode(ctx: *mut mldsa_ctx, set: c_int, c_tilde: *const u8, z: *const i32, h: *const i32, sigs: *mut u8, ok: *mut u8,
        n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_sig_decode(ctx: *mut mldsa_ctx, set: c_int, sigs: *const u8, c_tilde: *mut u8, z: *mut i32, h: *mut i32, ok: *mut u8, n_ops: usize,
        stream: *mut c_void) -> c_int;
    pub fn mldsa_w1_encode(ctx: *mut mldsa_ctx, set: c_int, w1: *const i32, out: *mut u8, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_mat_vec_mul(ctx: *mut mldsa_ctx, set: c_int, a_hat: *const i32, u_hat: *const i32, w_hat: *mut i32, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_pointwise_mont(ctx: *mut mldsa_ctx, c_hat: *const i32, v_hat_mont: *const i32, out: *mut i32, polys_per_op: usize, n_ops: usize,
        stream: *mut c_void) -> c_int;
    pub fn mldsa_add_vector_ntt(ctx: *mut mldsa_ctx, a: *const i32, b: *const i32, out: *mut i32, n_polys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_infinity_norm(ctx: *mut mldsa_ctx, polys: *const i32, polys_per_op: usize, n_ops: usize, norms: *mut i32, stream: *mut c_void) -> c_int;
    pub fn mldsa_verify_arith(ctx: *mut mldsa_ctx, set: c_int, a_hat: *const i32, z: *const i32, c: *const i32, t1_d2_hat_mont: *const i32,
        w_out: *mut i32, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_expand_a(ctx: *mut mldsa_ctx, set: c_int, rho: *const u8, a_hat: *mut i32, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_expand_s(ctx: *mut mldsa_ctx, set: c_int, rho_prime: *const u8, s1s2: *mut i32, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_expand_mask(ctx: *mut mldsa_ctx, set: c_int, rho_pp: *const u8, kappa: *const u16, y: *mut i32, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_sample_in_ball(ctx: *mut mldsa_ctx, set: c_int, c_tilde: *const u8, c: *mut i32, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_verify(ctx: *mut mldsa_ctx, set: c_int, mode: c_int, rho: *const u8, tr: *const u8, t1_d2_hat_mont: *const i32, n_keys: usize,
        key_idx: *const u32, msgs: *const u8, msg_off: *const u64, ctxs: *const u8, ctx_off: *const u64, sigs: *const u8, ok: *mut u8, n_ops: usize,
        stream: *mut c_void) -> c_int;
    pub fn mldsa_verify_pk(ctx: *mut mldsa_ctx, set: c_int, mode: c_int, pk: *const u8, n_keys: usize, key_idx: *const u32, msgs: *const u8,
        msg_off: *const u64, ctxs: *const u8, ctx_off: *const u64, sigs: *const u8, ok: *mut u8, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_check_offsets(off: *const u64, n_ops: usize) -> c_int;
    pub fn mldsa_pk_expand(ctx: *mut mldsa_ctx, set: c_int, pk: *const u8, rho: *mut u8, tr: *mut u8, t1_d2_hat_mont: *mut i32, n_keys: usize,
        stream: *mut c_void) -> c_int;
    pub fn mldsa_sk_expand(ctx: *mut mldsa_ctx, set: c_int, sk: *const u8, rho: *mut u8, cap_k: *mut u8, tr: *mut u8, s_1_hat_mont: *mut i32,
        s_2_hat_mont: *mut i32, t_0_hat_mont: *mut i32, n_keys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_pk_into_bytes(ctx: *mut mldsa_ctx, set: c_int, rho: *const u8, t1_d2_hat_mont: *const i32, pk: *mut u8, n_keys: usize,
        stream: *mut c_void) -> c_int;
    pub fn mldsa_sk_into_bytes(ctx: *mut mldsa_ctx, set: c_int, rho: *const u8, cap_k: *const u8, tr: *const u8, s_1_hat_mont: *const i32,
        s_2_hat_mont: *const i32, t_0_hat_mont: *const i32, sk: *mut u8, n_keys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_get_public_key(ctx: *mut mldsa_ctx, set: c_int, rho: *const u8, tr: *const u8, s_1_hat_mont: *const i32, s_2_hat_mont: *const i32,
        pk_rho: *mut u8, pk_tr: *mut u8, pk_t1_d2_hat_mont: *mut i32, n_keys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_keygen(ctx: *mut mldsa_ctx, set: c_int, xi: *const u8, pk: *mut u8, sk: *mut u8, n_keys: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_sign(ctx: *mut mldsa_ctx, set: c_int, mode: c_int, rho: *const u8, cap_k: *const u8, tr: *const u8, s_1_hat_mont: *const i32,
        s_2_hat_mont: *const i32, t_0_hat_mont: *const i32, n_keys: usize, key_idx: *const u32, msgs: *const u8, msg_off: *const u64,
        ctxs: *const u8, ctx_off: *const u64, rnd: *const u8, sigs: *mut u8, status: *mut i32, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_sign_async(ctx: *mut mldsa_ctx, set: c_int, mode: c_int, rho: *const u8, cap_k: *const u8, tr: *const u8, s_1_hat_mont: *const i32,
        s_2_hat_mont: *const i32, t_0_hat_mont: *const i32, n_keys: usize, key_idx: *const u32, msgs: *const u8, msg_off: *const u64,
        ctxs: *const u8, ctx_off: *const u64, rnd: *const u8, sigs: *mut u8, status: *mut i32, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_verify_cached_a(ctx: *mut mldsa_ctx, set: c_int, mode: c_int, a_hat: *const i32, tr: *const u8, t1_d2_hat_mont: *const i32,
        n_keys: usize, key_idx: *const u32, msgs: *const u8, msg_off: *const u64, ctxs: *const u8, ctx_off: *const u64, sigs: *const u8, ok: *mut u8,
        n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_sign_cached_a(ctx: *mut mldsa_ctx, set: c_int, mode: c_int, a_hat: *const i32, cap_k: *const u8, tr: *const u8,
        s_1_hat_mont: *const i32, s_2_hat_mont: *const i32, t_0_hat_mont: *const i32, n_keys: usize, key_idx: *const u32, msgs: *const u8,
        msg_off: *const u64, ctxs: *const u8, ctx_off: *const u64, rnd: *const u8, sigs: *mut u8, status: *mut i32, n_ops: usize, stream: *mut c_void) -> c_int;
    pub fn mldsa_verify_host(ctx: *mut mldsa_ctx, set: c_int, mode: c_int, pk: *const u8, n_keys: usize, key_idx: *const u32, msgs: *const u8,
        msg_off: *const u64, ctxs: *const u8, ctx_off: *const u64, sigs: *const u8, ok: *mut u8, n_ops: usize) -> c_int;
    pub fn mldsa_sign_host(ctx: *mut mldsa_ctx, set: c_int, mode: c_int, sk: *const u8, n_keys: usize, key_idx: *const u32, msgs: *const u8,
        msg_off: *const u64, ctxs: *const u8, ctx_off: *const u64, rnd: *const u8, sigs: *mut u8, status: *mut i32, n_ops: usize) -> c_int;
    pub fn mldsa_keygen_host(ctx: *mut mldsa_ctx, set: c_int, xi: *const u8, pk: *mut u8, sk: *mut u8, n_keys: usize) -> c_int;
    pub fn mldsa_group_create(device_ids: *const c_int, n: c_int, out: *mut *mut mldsa_group) -> c_int;
    pub fn mldsa_group_destroy(g: *mut mldsa_group);
    pub fn mldsa_group_size(g: *const mldsa_group) -> c_int;
    pub fn mldsa_group_ctx(g: *mut mldsa_group, i: c_int) -> *mut mldsa_ctx;
    pub fn mldsa_group_shard(n_ops: usize, n_parts: c_int, part: c_int, first: *mut usize, count: *mut usize) -> c_int;
    pub fn mldsa_verify_host_group(g: *mut mldsa_group, set: c_int, mode: c_int, pk: *const u8, n_keys: usize, key_idx: *const u32, msgs: *const u8,
        msg_off: *const u64, ctxs: *const u8, ctx_off: *const u64, sigs: *const u8, ok: *mut u8, n_ops: usize) -> c_int;
    pub fn mldsa_sign_host_group(g: *mut mldsa_group, set: c_int, mode: c_int, sk: *const u8, n_keys: usize, key_idx: *const u32, msgs: *const u8,
        msg_off: *const u64, ctxs: *const u8, ctx_off: *const u64, rnd: *const u8, sigs: *mut u8, status: *mut i32, n_ops: usize) -> c_int;
    pub fn mldsa_keygen_host_group(g: *mut mldsa_group, set: c_int, xi: *const u8, pk: *mut u8, sk: *mut u8, n_keys: usize) -> c_int;
    pub fn mldsa_verify_group(g: *mut mldsa_group, set: c_int, mode: c_int, slices: *const mldsa_verify_slice, wait: c_int) -> c_int;
    pub fn mldsa_sign_group(g: *mut mldsa_group, set: c_int, mode: c_int, slices: *const mldsa_sign_slice, wait: c_int) -> c_int;
    pub fn mldsa_keygen_group(g: *mut mldsa_group, set: c_int, slices: *const mldsa_keygen_slice, wait: c_int) -> c_int;
    pub fn mldsa_group_sync(g: *mut mldsa_group) -> c_int;
    pub fn mldsa_group_allgather(g: *mut mldsa_group, bufs: *const *mut u8, n_ops: usize, use_rccl: c_int) -> c_int;
    pub fn mldsa_group_rccl_info(g: *const mldsa_group, buf: *mut c_char, buf_len: usize) -> c_int;
    pub fn mldsa_batcher_create(ctx: *mut mldsa_ctx, set: c_int, max_batch: usize, max_wait_us: c_uint, cache_keys: usize,
        out: *mut *mut mldsa_batcher) -> c_int;
    pub fn mldsa_batcher_create_on(device_ids: *const c_int, n: c_int, set: c_int, max_batch: usize, max_wait_us: c_uint, cache_keys: usize,
        out: *mut *mut mldsa_batcher) -> c_int;
    pub fn mldsa_batcher_lanes(b: *const mldsa_batcher) -> c_int;
    pub fn mldsa_batcher_destroy(b: *mut mldsa_batcher);
    pub fn mldsa_batcher_verify(b: *mut mldsa_batcher, mode: c_int, pk: *const u8, msg: *const u8, msg_len: usize, ctx: *const u8, ctx_len: usize,
        sig: *const u8, ok: *mut u8) -> c_int;
    pub fn mldsa_batcher_sign(b: *mut mldsa_batcher, mode: c_int, sk: *const u8, msg: *const u8, msg_len: usize, ctx: *const u8, ctx_len: usize,
        rnd: *const u8, sig: *mut u8) -> c_int;
    pub fn mldsa_batcher_keygen(b: *mut mldsa_batcher, xi: *const u8, pk: *mut u8, sk: *mut u8) -> c_int;
    pub fn mldsa_batcher_get_stats(b: *mut mldsa_batcher, out: *mut mldsa_batcher_stats) -> c_int;
    pub fn mldsa_batcher_forget_key(b: *mut mldsa_batcher, key: *const u8, key_len: usize) -> c_int;
    pub fn mldsa_batcher_flush_keys(b: *mut mldsa_batcher) -> c_int;
    pub fn mldsa_batcher_set_private_key_cache(b: *mut mldsa_batcher, on: c_int) -> c_int;
}
