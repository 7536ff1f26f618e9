"""Host-side mirror of the reference's crate-private hot-path seams, batched, on the GPU.

Same names, argument meaning and error behaviour as the reference functions they stand
for (src/ntt.rs, src/helpers.rs, src/hashing.rs); every call goes through the C ABI of
include/mldsa_hip.h.  PyTorch is used only as device-memory / stream plumbing: arguments
and results are int32 / uint8 CUDA tensors whose data_ptr() is handed to the library.
"""
import ctypes as C

import torch

from . import _lib

N = 256
Q = 8380417


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _stream(device=None):
    """torch's current stream ON `device` (None: the current device) as a hipStream_t.  Every call of a context must name a stream
    of the context's own device: with several GPUs in one process the current device is not necessarily that one."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _bytes(t, name, row):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
        raise TypeError(f"{name}: expected a contiguous uint8 CUDA tensor")
    if row and t.numel() % row:
        raise ValueError(f"{name}: length is not a multiple of {row}")
    return t


def _polys(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise TypeError(f"{name}: expected a contiguous int32 CUDA tensor")
    if t.numel() % N:
        raise ValueError(f"{name}: number of coefficients is not a multiple of 256")
    return t


class HotPath:
    """Owns one mldsa_ctx (device tables + workspaces) on `device`."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("fips204_amd needs a HIP device; there is no CPU fallback")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        h = C.c_void_p()
        _lib.check(self.lib.mldsa_ctx_create(device, C.byref(h)))
        self._h = h

    @classmethod
    def from_handle(cls, handle, device):
        """A non-owning view of an existing mldsa_ctx (e.g. mldsa_group_ctx(g, i)): close() leaves it alone."""
        self = cls.__new__(cls)
        self.lib = _lib.load()
        self.device = torch.device("cuda", device)
        self._h = handle
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "_h", None):
            if not getattr(self, "_borrowed", False):
                self.lib.mldsa_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- context knobs -----------------------------------------------------------
    def set_option(self, option, value):
        _lib.check(self.lib.mldsa_set_option(self._h, option, value))

    def get_option(self, option):
        return int(self.lib.mldsa_get_option(self._h, option))

    def stats(self):
        st = _lib.Stats()
        _lib.check(self.lib.mldsa_get_stats(self._h, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in _lib.Stats._fields_}

    def set_workspace(self, buf):
        """mldsa_ctx_set_workspace: a uint8 CUDA tensor of the caller's as the context's workspace (None: context-owned again).
        The caller keeps the tensor alive for as long as the context uses it."""
        if buf is None:
            _lib.check(self.lib.mldsa_ctx_set_workspace(self._h, None, 0))
        else:
            _bytes(buf, "workspace", 0)
            _lib.check(self.lib.mldsa_ctx_set_workspace(self._h, _ptr(buf), buf.numel()))
        self._ws_keepalive = buf

    def secret_residue(self):
        """mldsa_debug_secret_residue: (bytes scanned, non-zero bytes) of the last call's secret-dependent workspace span and
        of the staging buffers that held secrets -- test support for the mirror of ZeroizeOnDrop (src/types.rs:19, 45)"""
        scanned, nonzero = C.c_size_t(), C.c_size_t()
        _lib.check(self.lib.mldsa_debug_secret_residue(self._h, C.byref(scanned), C.byref(nonzero)))
        return scanned.value, nonzero.value

    def reserve(self, pset, op, n_ops):
        """Size the workspace ahead of time (mldsa_reserve): later calls of that size never wait for the device."""
        _lib.check(self.lib.mldsa_reserve(self._h, pset, op, n_ops))

    # ---- per-stage timing (HIP events on the launch stream) -----------------------
    def profile_enable(self, on=True):
        _lib.check(self.lib.mldsa_profile_enable(self._h, 1 if on else 0))

    def profile_report(self):
        import json
        buf = C.create_string_buffer(8192)
        _lib.check(self.lib.mldsa_profile_report(self._h, buf, len(buf)))
        return json.loads(buf.value.decode())

    # ---- src/ntt.rs ---------------------------------------------------------------
    def ntt(self, w, out=None):
        """ntt::<KL>(&[R; KL]) -> [T; KL]   (src/ntt.rs:14)"""
        w = _polys(w, "w")
        out = torch.empty_like(w) if out is None else out
        _lib.check(self.lib.mldsa_ntt(self._h, _ptr(w), _ptr(out), w.numel() // N, _stream(self.device)))
        return out

    def inv_ntt(self, w_hat, out=None):
        """inv_ntt::<KL>(&[T; KL]) -> [R; KL]   (src/ntt.rs:85)"""
        w_hat = _polys(w_hat, "w_hat")
        out = torch.empty_like(w_hat) if out is None else out
        _lib.check(self.lib.mldsa_inv_ntt(self._h, _ptr(w_hat), _ptr(out), w_hat.numel() // N, _stream(self.device)))
        return out

    # ---- src/helpers.rs -----------------------------------------------------------
    def to_mont(self, v):
        """to_mont(&[T; L]) -> [T; L]   (src/helpers.rs:131)"""
        v = _polys(v, "v")
        out = torch.empty_like(v)
        _lib.check(self.lib.mldsa_to_mont(self._h, _ptr(v), _ptr(out), v.numel() // N, _stream(self.device)))
        return out

    def _reduce(self, kind, w):
        w = _polys(w, "w")
        out = torch.empty_like(w)
        _lib.check(self.lib.mldsa_reduce(self._h, kind, _ptr(w), _ptr(out), w.numel() // N, _stream(self.device)))
        return out

    def partial_reduce32(self, w):
        """partial_reduce32, element-wise (src/helpers.rs:61-67): (-q, q)"""
        return self._reduce(0, w)

    def full_reduce32(self, w):
        """full_reduce32, element-wise (src/helpers.rs:70-76): [0, q)"""
        return self._reduce(1, w)

    def center_mod(self, w):
        """center_mod, element-wise (src/helpers.rs:88-95): (-q/2, q/2]"""
        return self._reduce(2, w)

    def rounding(self, pset, op, a, b=None):
        """high_low.rs element-wise: op 0 power2round -> (r1, r0), 1 decompose -> (r1, r0), 2 high_bits, 3 low_bits,
        4 make_hint(z = a, r = b), 5 use_hint(h = a, r = b)"""
        a = _polys(a, "a")
        out1 = torch.empty_like(a)
        out2 = torch.empty_like(a) if op <= 1 else None
        _lib.check(self.lib.mldsa_rounding(self._h, pset, op, _ptr(a), _ptr(_polys(b, "b")) if b is not None else None, _ptr(out1),
                                           _ptr(out2) if out2 is not None else None, a.numel() // N, _stream(self.device)))
        return (out1, out2) if op <= 1 else out1

    # ---- codecs as seams (conversion.rs, encodings.rs); every `ok` is a uint8 tensor, 1 = the reference returns Ok
    def xof(self, bits, data, off, out_len):
        """h256_xof (bits = 256) / g128_xof (bits = 128) (src/hashing.rs:13-27): data uint8, off int64 / uint64 [n_ops + 1] on the device
        -> (out uint8 [n_ops, out_len], bad uint8 [n_ops])"""
        n = off.numel() - 1
        out = torch.empty((n, out_len), dtype=torch.uint8, device=off.device)
        bad = torch.empty(n, dtype=torch.uint8, device=off.device)
        _lib.check(self.lib.mldsa_xof(self._h, bits, _ptr(data) if data is not None and data.numel() else None, _ptr(off), _ptr(out), out_len,
                                      _ptr(bad), n, _stream(self.device)))
        return out, bad

    def bit_pack(self, w, a, b):
        """bit_pack(w, a, b) (src/conversion.rs:143-186; a = 0: simple_bit_pack) -> uint8 [n_polys, 32 * bitlen(a + b)]"""
        w = _polys(w, "w")
        n = w.numel() // N
        out = torch.empty((n, 32 * int(a + b).bit_length()), dtype=torch.uint8, device=w.device)
        _lib.check(self.lib.mldsa_bit_pack(self._h, _ptr(w), a, b, _ptr(out), n, _stream(self.device)))
        return out

    def bit_unpack(self, v, a, b):
        """bit_unpack(v, a, b) (src/conversion.rs:227-262; a = 0: simple_bit_unpack) -> (w int32 [n_polys, 256], ok)"""
        row = 32 * int(a + b).bit_length()
        v = _bytes(v, "v", row)
        n = v.numel() // row
        w = torch.empty((n, N), dtype=torch.int32, device=v.device)
        ok = torch.empty(n, dtype=torch.uint8, device=v.device)
        _lib.check(self.lib.mldsa_bit_unpack(self._h, _ptr(v), a, b, _ptr(w), _ptr(ok), n, _stream(self.device)))
        return w, ok

    def hint_bit_pack(self, pset, h):
        """hint_bit_pack (src/conversion.rs:277-328): h int32 [n_ops, K, 256] -> (y uint8 [n_ops, omega + K], ok)"""
        p = _lib.get_params(pset)
        h = _polys(h, "h")
        n = h.numel() // (p.k * N)
        y = torch.empty((n, p.omega + p.k), dtype=torch.uint8, device=h.device)
        ok = torch.empty(n, dtype=torch.uint8, device=h.device)
        _lib.check(self.lib.mldsa_hint_bit_pack(self._h, pset, _ptr(h), _ptr(y), _ptr(ok), n, _stream(self.device)))
        return y, ok

    def hint_bit_unpack(self, pset, y):
        """hint_bit_unpack (src/conversion.rs:340-414): y uint8 [n_ops, omega + K] -> (h int32 [n_ops, K, 256], ok)"""
        p = _lib.get_params(pset)
        y = _bytes(y, "y", p.omega + p.k)
        n = y.numel() // (p.omega + p.k)
        h = torch.empty((n, p.k, N), dtype=torch.int32, device=y.device)
        ok = torch.empty(n, dtype=torch.uint8, device=y.device)
        _lib.check(self.lib.mldsa_hint_bit_unpack(self._h, pset, _ptr(y), _ptr(h), _ptr(ok), n, _stream(self.device)))
        return h, ok

    def sig_encode(self, pset, c_tilde, z, h):
        """sig_encode (src/encodings.rs:238-280) -> (sigs uint8 [n_ops, sig_len], ok)"""
        p = _lib.get_params(pset)
        c_tilde, z, h = _bytes(c_tilde, "c_tilde", p.ctilde_len), _polys(z, "z"), _polys(h, "h")
        n = c_tilde.numel() // p.ctilde_len
        if z.numel() != n * p.l * N or h.numel() != n * p.k * N:
            raise ValueError("sig_encode: z / h do not match the number of c_tilde rows")
        sigs = torch.empty((n, p.sig_len), dtype=torch.uint8, device=z.device)
        ok = torch.empty(n, dtype=torch.uint8, device=z.device)
        _lib.check(self.lib.mldsa_sig_encode(self._h, pset, _ptr(c_tilde), _ptr(z), _ptr(h), _ptr(sigs), _ptr(ok), n, _stream(self.device)))
        return sigs, ok

    def sig_decode(self, pset, sigs):
        """sig_decode (src/encodings.rs:290-328) -> (c_tilde, z int32 [n_ops, L, 256], h int32 [n_ops, K, 256], ok)"""
        p = _lib.get_params(pset)
        sigs = _bytes(sigs, "sigs", p.sig_len)
        n = sigs.numel() // p.sig_len
        dev = sigs.device
        c_tilde = torch.empty((n, p.ctilde_len), dtype=torch.uint8, device=dev)
        z = torch.empty((n, p.l, N), dtype=torch.int32, device=dev)
        h = torch.empty((n, p.k, N), dtype=torch.int32, device=dev)
        ok = torch.empty(n, dtype=torch.uint8, device=dev)
        _lib.check(self.lib.mldsa_sig_decode(self._h, pset, _ptr(sigs), _ptr(c_tilde), _ptr(z), _ptr(h), _ptr(ok), n, _stream(self.device)))
        return c_tilde, z, h, ok

    def w1_encode(self, pset, w1):
        """w1_encode (src/encodings.rs:338-360): w1 int32 [n_ops, K, 256] -> uint8 [n_ops, w1_len]"""
        p = _lib.get_params(pset)
        w1 = _polys(w1, "w1")
        n = w1.numel() // (p.k * N)
        out = torch.empty((n, p.w1_len), dtype=torch.uint8, device=w1.device)
        _lib.check(self.lib.mldsa_w1_encode(self._h, pset, _ptr(w1), _ptr(out), n, _stream(self.device)))
        return out

    def mat_vec_mul(self, pset, a_hat, u_hat):
        """mat_vec_mul::<K, L>(&[[T; L]; K], &[T; L]) -> [T; K], batched over ops (src/helpers.rs:100)"""
        p = _lib.get_params(pset)
        a_hat, u_hat = _polys(a_hat, "a_hat"), _polys(u_hat, "u_hat")
        n_ops = u_hat.numel() // (p.l * N)
        if u_hat.numel() != n_ops * p.l * N or a_hat.numel() != n_ops * p.k * p.l * N:
            raise ValueError("mat_vec_mul: shape mismatch")
        out = torch.empty((n_ops, p.k, N), dtype=torch.int32, device=a_hat.device)
        _lib.check(self.lib.mldsa_mat_vec_mul(self._h, pset, _ptr(a_hat), _ptr(u_hat), _ptr(out), n_ops, _stream(self.device)))
        return out

    def pointwise_mont(self, c_hat, v_hat_mont):
        """c_hat o v_hat_mont, inlined at src/ml_dsa.rs:243-260, 288-295"""
        c_hat, v = _polys(c_hat, "c_hat"), _polys(v_hat_mont, "v_hat_mont")
        n_ops = c_hat.numel() // N
        if n_ops == 0 or v.numel() % (n_ops * N):
            raise ValueError("pointwise_mont: shape mismatch")
        ppo = v.numel() // (n_ops * N)
        out = torch.empty_like(v)
        _lib.check(self.lib.mldsa_pointwise_mont(self._h, _ptr(c_hat), _ptr(v), _ptr(out), ppo, n_ops, _stream(self.device)))
        return out

    def add_vector_ntt(self, a, b):
        """add_vector_ntt (src/helpers.rs:125)"""
        a, b = _polys(a, "a"), _polys(b, "b")
        if a.numel() != b.numel():
            raise ValueError("add_vector_ntt: shape mismatch")
        out = torch.empty_like(a)
        _lib.check(self.lib.mldsa_add_vector_ntt(self._h, _ptr(a), _ptr(b), _ptr(out), a.numel() // N, _stream(self.device)))
        return out

    def infinity_norm(self, w, polys_per_op):
        """infinity_norm::<ROW>(&[R; ROW]) -> i32 per op (src/helpers.rs:138)"""
        w = _polys(w, "w")
        n_ops = w.numel() // (polys_per_op * N)
        out = torch.empty(n_ops, dtype=torch.int32, device=w.device)
        _lib.check(self.lib.mldsa_infinity_norm(self._h, _ptr(w), polys_per_op, n_ops, _ptr(out), _stream(self.device)))
        return out

    def verify_arith(self, pset, a_hat, z, c, t1_d2_hat_mont, out=None):
        """w' = inv_ntt(A_hat * ntt(z) - ntt(c) o t1_d2_hat_mont)   (src/ml_dsa.rs:407-416)"""
        p = _lib.get_params(pset)
        for t, nm in ((a_hat, "a_hat"), (z, "z"), (c, "c"), (t1_d2_hat_mont, "t1_d2_hat_mont")):
            _polys(t, nm)
        n_ops = c.numel() // N
        if (a_hat.numel() != n_ops * p.k * p.l * N or z.numel() != n_ops * p.l * N
                or t1_d2_hat_mont.numel() != n_ops * p.k * N):
            raise ValueError("verify_arith: shape mismatch")
        if out is None:
            out = torch.empty((n_ops, p.k, N), dtype=torch.int32, device=c.device)
        _lib.check(self.lib.mldsa_verify_arith(self._h, pset, _ptr(a_hat), _ptr(z), _ptr(c),
                                               _ptr(t1_d2_hat_mont), _ptr(out), n_ops, _stream(self.device)))
        return out

    # ---- src/hashing.rs -----------------------------------------------------------
    def expand_a(self, pset, rho):
        """expand_a::<K, L>(&[u8; 32]) -> [[T; L]; K], one rho per op (src/hashing.rs:225)"""
        p = _lib.get_params(pset)
        rho = _bytes(rho, "rho", 32)
        n_ops = rho.numel() // 32
        out = torch.empty((n_ops, p.k, p.l, N), dtype=torch.int32, device=rho.device)
        _lib.check(self.lib.mldsa_expand_a(self._h, pset, _ptr(rho), _ptr(out), n_ops, _stream(self.device)))
        return out

    def expand_s(self, pset, rho_prime):
        """expand_s::<K, L>(eta, &[u8; 64]) -> ([R; L], [R; K]) per op (src/hashing.rs:252)"""
        p = _lib.get_params(pset)
        rho_prime = _bytes(rho_prime, "rho_prime", 64)
        n_ops = rho_prime.numel() // 64
        out = torch.empty((n_ops, p.l + p.k, N), dtype=torch.int32, device=rho_prime.device)
        _lib.check(self.lib.mldsa_expand_s(self._h, pset, _ptr(rho_prime), _ptr(out), n_ops, _stream(self.device)))
        return out[:, :p.l], out[:, p.l:]

    def expand_mask(self, pset, rho_pp, kappa):
        """expand_mask::<L>(gamma1, &[u8; 64], mu: u16) -> [R; L] per op (src/hashing.rs:281)"""
        p = _lib.get_params(pset)
        rho_pp = _bytes(rho_pp, "rho_pp", 64)
        n_ops = rho_pp.numel() // 64
        if not (kappa.is_cuda and kappa.dtype == torch.int16 and kappa.numel() == n_ops):
            raise TypeError("kappa: expected an int16 CUDA tensor (u16 bit pattern) with one entry per op")
        out = torch.empty((n_ops, p.l, N), dtype=torch.int32, device=rho_pp.device)
        _lib.check(self.lib.mldsa_expand_mask(self._h, pset, _ptr(rho_pp), _ptr(kappa), _ptr(out), n_ops, _stream(self.device)))
        return out

    def sample_in_ball(self, pset, c_tilde):
        """sample_in_ball(tau, &c_tilde) -> R per op (src/hashing.rs:43)"""
        p = _lib.get_params(pset)
        c_tilde = _bytes(c_tilde, "c_tilde", p.ctilde_len)
        n_ops = c_tilde.numel() // p.ctilde_len
        out = torch.empty((n_ops, N), dtype=torch.int32, device=c_tilde.device)
        _lib.check(self.lib.mldsa_sample_in_ball(self._h, pset, _ptr(c_tilde), _ptr(out), n_ops, _stream(self.device)))
        return out
