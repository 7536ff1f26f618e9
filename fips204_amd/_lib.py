"""ctypes loader for the C-ABI library (include/mldsa_hip.h).

There is deliberately no fallback: if the HIP library is missing or a call fails the
caller gets an exception.  Nothing here imports or calls oracle/.
"""
import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmldsa_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "mldsa_hip.h")

OK = 0


class MldsaError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mldsa_hip error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "set", "k", "l", "eta", "tau", "lambda_", "gamma1", "gamma2", "omega", "beta",
        "ctilde_len", "pk_len", "sk_len", "sig_len", "w1_len")]


_P, _SZ, _I = C.c_void_p, C.c_size_t, C.c_int


class VerifySlice(C.Structure):
    """mldsa_verify_slice: the arguments of mldsa_verify for one device's slice (device pointers of THAT device)"""
    _fields_ = [("rho", _P), ("tr", _P), ("t1_d2_hat_mont", _P), ("n_keys", _SZ), ("key_idx", _P), ("msgs", _P), ("msg_off", _P),
                ("ctxs", _P), ("ctx_off", _P), ("sigs", _P), ("ok", _P), ("n_ops", _SZ), ("stream", _P)]


class SignSlice(C.Structure):
    """mldsa_sign_slice"""
    _fields_ = [("rho", _P), ("cap_k", _P), ("tr", _P), ("s_1_hat_mont", _P), ("s_2_hat_mont", _P), ("t_0_hat_mont", _P), ("n_keys", _SZ),
                ("key_idx", _P), ("msgs", _P), ("msg_off", _P), ("ctxs", _P), ("ctx_off", _P), ("rnd", _P), ("sigs", _P), ("status", _P),
                ("n_ops", _SZ), ("stream", _P)]


class KeygenSlice(C.Structure):
    """mldsa_keygen_slice"""
    _fields_ = [("xi", _P), ("pk", _P), ("sk", _P), ("n_keys", _SZ), ("stream", _P)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in (
        "graphs_captured", "graph_replays", "direct_calls", "workspace_growths", "sign_extra_rounds", "workspace_shrinks")]


class BatcherStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("batches", "requests", "largest_batch", "keys_expanded", "key_hits")]


OP_KEYGEN, OP_SIGN, OP_VERIFY = 1, 2, 3
OPT_GRAPHS, OPT_SPEC_TARGET, OPT_SPEC_MAX, OPT_VA_BLOCKS_PER_CU, OPT_GRAPH_CACHE, OPT_SIGN_ROUNDS = 1, 2, 3, 4, 5, 6
OPT_SIGN_LANES, OPT_SIGN_CT0_EXACT, OPT_SIGN_ASYNC_EXP, OPT_SIGN_LOOKAHEAD, OPT_WORKSPACE_CAP_MB = 7, 8, 9, 10, 11
OPT_COOP_HASH = 12
OPT_SMALL_FUSED = 13
ABI_VERSION = 6
ERR_PARAM, ERR_CTX_LEN, ERR_DEVICE, ERR_NOMEM, ERR_AGAIN = -1, -2, -3, -4, -5
ROUND_POWER2ROUND, ROUND_DECOMPOSE, ROUND_HIGH_BITS, ROUND_LOW_BITS, ROUND_MAKE_HINT, ROUND_USE_HINT = range(6)

# name -> argtypes (all return int unless listed in _RESTYPES)
_SIGNATURES = {
    "mldsa_ctx_create": [_I, C.POINTER(_P)],
    "mldsa_ctx_destroy": [_P],
    "mldsa_last_error": [],
    "mldsa_get_params": [_I, C.POINTER(Params)],
    "mldsa_device_count": [],
    "mldsa_abi_version": [],
    "mldsa_check_offsets": [_P, _SZ],
    "mldsa_get_stats_sized": [_P, _P, _SZ],
    "mldsa_debug_secret_residue": [_P, C.POINTER(_SZ), C.POINTER(_SZ)],
    "mldsa_debug_count_nonzero": [_P, _SZ, C.POINTER(_SZ)],
    "mldsa_verify_group": [_P, _I, _I, _P, _I],
    "mldsa_sign_group": [_P, _I, _I, _P, _I],
    "mldsa_keygen_group": [_P, _I, _P, _I],
    "mldsa_group_sync": [_P],
    "mldsa_batcher_destroy": [_P],
    "mldsa_batcher_create": [_P, _I, _SZ, C.c_uint, _SZ, _P],
    "mldsa_batcher_create_on": [_P, _I, _I, _SZ, C.c_uint, _SZ, _P],
    "mldsa_batcher_lanes": [_P],
    "mldsa_batcher_verify": [_P, _I, _P, _P, _SZ, _P, _SZ, _P, _P],
    "mldsa_batcher_sign": [_P, _I, _P, _P, _SZ, _P, _SZ, _P, _P],
    "mldsa_batcher_keygen": [_P, _P, _P, _P],
    "mldsa_batcher_get_stats": [_P, _P],
    "mldsa_batcher_forget_key": [_P, _P, _SZ],
    "mldsa_batcher_flush_keys": [_P],
    "mldsa_batcher_set_private_key_cache": [_P, _I],
    "mldsa_ctx_device": [_P],
    "mldsa_reserve": [_P, _I, _I, _SZ],
    "mldsa_ctx_set_workspace": [_P, _P, _SZ],
    "mldsa_set_option": [_P, _I, C.c_long],
    "mldsa_get_option": [_P, _I],
    "mldsa_get_stats": [_P, C.POINTER(Stats)],
    "mldsa_profile_enable": [_P, _I],
    "mldsa_profile_report": [_P, C.c_char_p, _SZ],
    "mldsa_malloc": [C.POINTER(_P), _SZ],
    "mldsa_ctx_malloc": [_P, C.POINTER(_P), _SZ],
    "mldsa_free": [_P],
    "mldsa_host_alloc": [C.POINTER(_P), _SZ],
    "mldsa_host_free": [_P],
    "mldsa_memcpy_h2d": [_P, _P, _SZ, _P],
    "mldsa_memcpy_d2h": [_P, _P, _SZ, _P],
    "mldsa_memset": [_P, _I, _SZ, _P],
    "mldsa_stream_sync": [_P],
    "mldsa_ntt": [_P, _P, _P, _SZ, _P],
    "mldsa_inv_ntt": [_P, _P, _P, _SZ, _P],
    "mldsa_to_mont": [_P, _P, _P, _SZ, _P],
    "mldsa_reduce": [_P, _I, _P, _P, _SZ, _P],
    "mldsa_rounding": [_P, _I, _I, _P, _P, _P, _P, _SZ, _P],
    "mldsa_xof": [_P, _I, _P, _P, _P, _SZ, _P, _SZ, _P],
    "mldsa_bit_pack": [_P, _P, _I, _I, _P, _SZ, _P],
    "mldsa_bit_unpack": [_P, _P, _I, _I, _P, _P, _SZ, _P],
    "mldsa_hint_bit_pack": [_P, _I, _P, _P, _P, _SZ, _P],
    "mldsa_hint_bit_unpack": [_P, _I, _P, _P, _P, _SZ, _P],
    "mldsa_sig_encode": [_P, _I, _P, _P, _P, _P, _P, _SZ, _P],
    "mldsa_sig_decode": [_P, _I, _P, _P, _P, _P, _P, _SZ, _P],
    "mldsa_w1_encode": [_P, _I, _P, _P, _SZ, _P],
    "mldsa_mat_vec_mul": [_P, _I, _P, _P, _P, _SZ, _P],
    "mldsa_pointwise_mont": [_P, _P, _P, _P, _SZ, _SZ, _P],
    "mldsa_add_vector_ntt": [_P, _P, _P, _P, _SZ, _P],
    "mldsa_infinity_norm": [_P, _P, _SZ, _SZ, _P, _P],
    "mldsa_verify_arith": [_P, _I, _P, _P, _P, _P, _P, _SZ, _P],
    "mldsa_expand_a": [_P, _I, _P, _P, _SZ, _P],
    "mldsa_expand_s": [_P, _I, _P, _P, _SZ, _P],
    "mldsa_expand_mask": [_P, _I, _P, _P, _P, _SZ, _P],
    "mldsa_sample_in_ball": [_P, _I, _P, _P, _SZ, _P],
    # ctx, set, mode, rho | a_hat, tr, t1, n_keys, key_idx, msgs, msg_off, ctxs, ctx_off, sigs, ok, n_ops, stream
    "mldsa_verify": [_P, _I, _I, _P, _P, _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _SZ, _P],
    "mldsa_verify_cached_a": [_P, _I, _I, _P, _P, _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _SZ, _P],
    # ctx, set, mode, pk (wire bytes), n_keys, key_idx, msgs, msg_off, ctxs, ctx_off, sigs, ok, n_ops, stream
    "mldsa_verify_pk": [_P, _I, _I, _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _SZ, _P],
    "mldsa_pk_expand": [_P, _I, _P, _P, _P, _P, _SZ, _P],
    "mldsa_sk_expand": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _SZ, _P],
    "mldsa_pk_into_bytes": [_P, _I, _P, _P, _P, _SZ, _P],
    "mldsa_sk_into_bytes": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _SZ, _P],
    "mldsa_get_public_key": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _SZ, _P],
    "mldsa_keygen": [_P, _I, _P, _P, _P, _SZ, _P],
    # ctx, set, mode, rho | a_hat, K, tr, s1, s2, t0, n_keys, key_idx, msgs, msg_off, ctxs, ctx_off, rnd, sigs, status, n_ops, stream
    "mldsa_sign": [_P, _I, _I] + [_P] * 6 + [_SZ] + [_P] * 8 + [_SZ, _P],
    "mldsa_sign_async": [_P, _I, _I] + [_P] * 6 + [_SZ] + [_P] * 8 + [_SZ, _P],
    "mldsa_sign_cached_a": [_P, _I, _I] + [_P] * 6 + [_SZ] + [_P] * 8 + [_SZ, _P],
    # host-memory variants: ctx, set, mode, keys, n_keys, key_idx, msgs, msg_off, ctxs, ctx_off, ...
    "mldsa_verify_host": [_P, _I, _I, _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _SZ],
    "mldsa_sign_host": [_P, _I, _I, _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _P, _SZ],
    "mldsa_keygen_host": [_P, _I, _P, _P, _P, _SZ],
    "mldsa_group_create": [C.POINTER(_I), _I, C.POINTER(_P)],
    "mldsa_group_destroy": [_P],
    "mldsa_group_size": [_P],
    "mldsa_group_ctx": [_P, _I],
    "mldsa_group_shard": [_SZ, _I, _I, C.POINTER(_SZ), C.POINTER(_SZ)],
    "mldsa_verify_host_group": [_P, _I, _I, _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _SZ],
    "mldsa_sign_host_group": [_P, _I, _I, _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _P, _SZ],
    "mldsa_keygen_host_group": [_P, _I, _P, _P, _P, _SZ],
    "mldsa_group_allgather": [_P, C.POINTER(_P), _SZ, _I],
    "mldsa_group_rccl_info": [_P, C.c_char_p, _SZ],
}
_RESTYPES = {"mldsa_ctx_destroy": None, "mldsa_last_error": C.c_char_p, "mldsa_get_option": C.c_long,
             "mldsa_group_destroy": None, "mldsa_group_ctx": _P, "mldsa_batcher_destroy": None}

_lib = None


def declared_symbols():
    """Every function name declared in include/mldsa_hip.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mldsa_[a-z0-9_]+)\s*\(", text)))


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m fips204_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    try:  # share one HIP runtime with PyTorch when both live in this process
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


def check(rc):
    if rc != OK:
        raise MldsaError(rc, load().mldsa_last_error().decode(errors="replace"))


def get_params(pset):
    p = Params()
    check(load().mldsa_get_params(pset, C.byref(p)))
    return p
