"""Batched mirror of the reference's public API for the accelerated path.

`ml_dsa_44`, `ml_dsa_65`, `ml_dsa_87` mirror the modules the `functionality!()` macro stamps
out (src/lib.rs:116-614): KeyGen / Signer / Verifier / SerDes (src/traits.rs), with every
method taking a *batch* of independent operations.  Keys live on the device in the
reference's expanded form (src/types.rs:19-41), field by field.  All compute goes through
the C ABI (include/mldsa_hip.h); there is no CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from .hotpath import HotPath, N, _ptr, _stream

MODE_PURE, MODE_INTERNAL, MODE_PREHASH = 0, 1, 2

# Ph (src/types.rs:5-12) and hash_message (src/hashing.rs:316-354): the pre-hash of HashML-DSA is message-length-bound
# host work; the device sees OID || PH(M) as the message of a MODE_PREHASH call.
PH_SHA256, PH_SHA512, PH_SHAKE128 = "SHA256", "SHA512", "SHAKE128"
_PH_OID = bytes([0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02])


class OsRng:
    """rand_core::OsRng as the reference's default generator (src/traits.rs:45, 157, 250): the kernel's CSPRNG"""

    def fill_bytes(self, n):
        import os
        return os.urandom(n)


def hash_message(message, ph):
    """OID || PH(M): DER object identifier of the hash (11 bytes) followed by its digest (32 / 64 / 32 bytes)."""
    import hashlib
    message = bytes(message)
    if ph == PH_SHA256:
        return _PH_OID + b"\x01" + hashlib.sha256(message).digest()
    if ph == PH_SHA512:
        return _PH_OID + b"\x03" + hashlib.sha512(message).digest()
    if ph == PH_SHAKE128:
        return _PH_OID + b"\x0b" + hashlib.shake_128(message).digest(32)
    raise ValueError("Ph: SHA256, SHA512 or SHAKE128")


def _cat_with_offsets(items, device):
    """list of bytes -> (uint8 device buffer, uint64 offsets[n + 1] on device)"""
    lens = np.fromiter((len(b) for b in items), dtype=np.uint64, count=len(items))
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    flat = b"".join(bytes(b) for b in items) or b"\0"
    buf = torch.frombuffer(bytearray(flat), dtype=torch.uint8).to(device)
    return buf, torch.from_numpy(off.view(np.int64)).to(device)


class PublicKeys:
    """n expanded public keys: PublicKey { rho, tr, t1_d2_hat_mont } (src/types.rs:35-41)."""

    def __init__(self, pset, rho, tr, t1_d2_hat_mont):
        self.pset, self.rho, self.tr, self.t1_d2_hat_mont = pset, rho, tr, t1_d2_hat_mont

    def __len__(self):
        return self.rho.shape[0]


class PrivateKeys:
    """n expanded private keys (src/types.rs:19-28).  Like the reference's `Zeroize, ZeroizeOnDrop` struct
    the secret fields are wiped when the object goes away (before their memory returns to the allocator)."""

    def __init__(self, pset, rho, cap_k, tr, s_1_hat_mont, s_2_hat_mont, t_0_hat_mont):
        self.pset, self.rho, self.cap_k, self.tr = pset, rho, cap_k, tr
        self.s_1_hat_mont, self.s_2_hat_mont, self.t_0_hat_mont = s_1_hat_mont, s_2_hat_mont, t_0_hat_mont

    def __len__(self):
        return self.rho.shape[0]

    def zeroize(self):
        for t in (self.cap_k, self.s_1_hat_mont, self.s_2_hat_mont, self.t_0_hat_mont):
            if t is not None:
                t.zero_()

    def __del__(self):
        try:
            self.zeroize()
        except Exception:
            pass


def _check_key_idx(key_idx, n_keys, n_ops):
    """Host-side courtesy check (the library itself also refuses out-of-range indices per op).
    A uint32 array is passed through without a copy."""
    if key_idx is None:
        if n_keys < n_ops:
            raise ValueError(f"{n_ops} operations but only {n_keys} keys and no key_idx")
        return None
    a = np.asarray(key_idx)
    if a.dtype != np.uint32:
        a64 = a.astype(np.int64)
        if n_ops and a64.size and a64.min() < 0:
            raise IndexError(f"key_idx out of range (n_keys = {n_keys})")
        a = a64.astype(np.uint32) if (not a64.size or a64.max() < 2 ** 32) else np.full(a64.shape, 0xFFFFFFFF, np.uint32)
    if a.shape != (n_ops,):
        raise ValueError("key_idx: one entry per operation expected")
    if n_ops and int(a.max()) >= n_keys:
        raise IndexError(f"key_idx out of range (n_keys = {n_keys})")
    return np.ascontiguousarray(a)


class MlDsa:
    """One parameter set (src/lib.rs:639-656 / 681-698 / 723-740) on one GPU."""

    def __init__(self, pset, device=0, hotpath=None):
        self.pset = pset
        self.hp = hotpath or HotPath(device)
        self.lib = self.hp.lib
        p = _lib.get_params(pset)
        self.params = p
        self.PK_LEN, self.SK_LEN, self.SIG_LEN = p.pk_len, p.sk_len, p.sig_len
        self.device = self.hp.device

    # ---- Verifier (src/traits.rs:330-362; src/lib.rs:364-411) -------------------------
    def verify(self, pks, messages, sigs, ctxs=None, key_idx=None, mode=MODE_PURE):
        """PublicKey::verify for a batch: returns a bool array, one entry per operation.

        `sigs`: uint8 CUDA tensor [n_ops, SIG_LEN] or a list of byte strings (a signature of
        the wrong length verifies as False, like a failed `try_into()` in the reference's
        callers).  `ctxs`: list of byte strings or None (= empty)."""
        n_ops = len(messages)
        wrong_len = None
        if not isinstance(sigs, torch.Tensor):
            wrong_len = np.array([len(s) != self.SIG_LEN for s in sigs], dtype=bool)
            flat = b"".join(bytes(s) if len(s) == self.SIG_LEN else bytes(self.SIG_LEN) for s in sigs)
            sigs = torch.frombuffer(bytearray(flat or b"\0"), dtype=torch.uint8).to(self.device)
        msg_buf, msg_off = _cat_with_offsets(messages, self.device)
        ctx_buf = ctx_off = None
        if ctxs is not None:
            ctx_buf, ctx_off = _cat_with_offsets(ctxs, self.device)
        if key_idx is None and len(pks) != n_ops:
            key_idx = np.arange(n_ops, dtype=np.uint32) % len(pks)
        key_idx = _check_key_idx(key_idx, len(pks), n_ops)
        kidx = None
        if key_idx is not None:
            kidx = torch.as_tensor(key_idx.view(np.int32)).to(self.device)
        ok = torch.zeros(max(n_ops, 1), dtype=torch.uint8, device=self.device)
        self.verify_device(pks, msg_buf, msg_off, sigs, ok, n_ops, ctx_buf, ctx_off, kidx, mode)
        torch.cuda.synchronize(self.device)
        res = ok[:n_ops].cpu().numpy().astype(bool)
        if wrong_len is not None:
            res &= ~wrong_len
        return res

    def hash_verify(self, pks, messages, sigs, ctxs=None, ph=PH_SHA512, key_idx=None):
        """PublicKey::hash_verify (src/traits.rs:361, src/lib.rs:391-411) for a batch: HashML-DSA.Verify with the
        pre-hash `ph` computed on the host."""
        return self.verify(pks, [hash_message(m, ph) for m in messages], sigs, ctxs=ctxs, key_idx=key_idx, mode=MODE_PREHASH)

    def expand_a_for_keys(self, keys):
        """A_hat = ExpandA(rho) of every key of a PublicKeys / PrivateKeys batch: the `cap_a_hat`
        pre-compute the reference lists as an open optimisation (benches/README.md:4-8).  Pass the result
        as `a_hat=` to verify_device / sign_device to skip the per-operation ExpandA."""
        return self.hp.expand_a(self.pset, keys.rho.contiguous())

    def verify_device(self, pks, msg_buf, msg_off, sigs, ok, n_ops, ctx_buf=None, ctx_off=None, key_idx=None,
                      mode=MODE_PURE, a_hat=None):
        """Same, everything already resident in HBM (what bench.py times)."""
        null = C.c_void_p(0)
        fn, first = (self.lib.mldsa_verify, pks.rho) if a_hat is None else (self.lib.mldsa_verify_cached_a, a_hat)
        _lib.check(fn(
            self.hp._h, self.pset, mode, _ptr(first), _ptr(pks.tr), _ptr(pks.t1_d2_hat_mont), len(pks),
            _ptr(key_idx) if key_idx is not None else null, _ptr(msg_buf), _ptr(msg_off),
            _ptr(ctx_buf) if ctx_buf is not None else null, _ptr(ctx_off) if ctx_off is not None else null,
            _ptr(sigs), _ptr(ok), n_ops, _stream(self.device)))
        return ok

    def verify_pk_device(self, pk_bytes, msg_buf, msg_off, sigs, ok, n_ops, ctx_buf=None, ctx_off=None, key_idx=None, mode=MODE_PURE):
        """mldsa_verify_pk: PublicKey::try_from_bytes + verify in one call -- pk_bytes = uint8 CUDA tensor [n_keys, PK_LEN] in wire format
        (key_idx None: op i uses key i).  Same verdicts as public_keys_from_bytes + verify_device."""
        null = C.c_void_p(0)
        pk = self._key_bytes(pk_bytes, self.PK_LEN, "pk")
        _lib.check(self.lib.mldsa_verify_pk(
            self.hp._h, self.pset, mode, _ptr(pk), pk.shape[0], _ptr(key_idx) if key_idx is not None else null, _ptr(msg_buf), _ptr(msg_off),
            _ptr(ctx_buf) if ctx_buf is not None else null, _ptr(ctx_off) if ctx_off is not None else null, _ptr(sigs), _ptr(ok), n_ops,
            _stream(self.device)))
        return ok

    def verify_pk(self, pk_bytes, messages, sigs, ctxs=None, key_idx=None, mode=MODE_PURE):
        """PublicKey::try_from_bytes(pk)?.verify(message, sig, ctx) for a batch of wire-format keys (src/lib.rs:471-475, 364-380)"""
        n_ops = len(messages)
        pk = self._key_bytes(pk_bytes, self.PK_LEN, "pk")
        msg_buf, msg_off = _cat_with_offsets(messages, self.device)
        ctx_buf = ctx_off = None
        if ctxs is not None:
            ctx_buf, ctx_off = _cat_with_offsets(ctxs, self.device)
        kidx = _check_key_idx(key_idx, pk.shape[0], n_ops)
        if kidx is not None:
            kidx = torch.as_tensor(kidx.view(np.int32)).to(self.device)
        sg = self._key_bytes(sigs, self.SIG_LEN, "sigs") if n_ops else torch.zeros((1, self.SIG_LEN), dtype=torch.uint8, device=self.device)
        ok = torch.zeros(max(n_ops, 1), dtype=torch.uint8, device=self.device)
        self.verify_pk_device(pk, msg_buf, msg_off, sg, ok, n_ops, ctx_buf, ctx_off, kidx, mode)
        torch.cuda.synchronize(self.device)
        return ok[:n_ops].cpu().numpy().astype(bool)

    # ---- SerDes (src/traits.rs:372-424; src/lib.rs:421-424, 471-475) ------------------
    def _key_bytes(self, keys, length, what):
        if isinstance(keys, torch.Tensor):
            if keys.dtype != torch.uint8 or keys.numel() % length:
                raise ValueError(f"{what}: expected uint8 tensor of n * {length} bytes")
            return keys.to(self.device).contiguous().view(-1, length)
        for b in keys:
            if len(b) != length:  # the reference's ByteArray is a fixed-size array type
                raise ValueError(f"{what}: wrong length {len(b)} (expected {length})")
        return torch.frombuffer(bytearray(b"".join(bytes(b) for b in keys)), dtype=torch.uint8).to(self.device).view(-1, length)

    def empty_public_keys(self, n):
        k = self.params.k
        return PublicKeys(self.pset, torch.empty((n, 32), dtype=torch.uint8, device=self.device),
                          torch.empty((n, 64), dtype=torch.uint8, device=self.device),
                          torch.empty((n, k, N), dtype=torch.int32, device=self.device))

    def empty_private_keys(self, n):
        k, l, dev = self.params.k, self.params.l, self.device
        return PrivateKeys(self.pset, torch.empty((n, 32), dtype=torch.uint8, device=dev), torch.empty((n, 32), dtype=torch.uint8, device=dev),
                           torch.empty((n, 64), dtype=torch.uint8, device=dev), torch.empty((n, l, N), dtype=torch.int32, device=dev),
                           torch.empty((n, k, N), dtype=torch.int32, device=dev), torch.empty((n, k, N), dtype=torch.int32, device=dev))

    def public_keys_from_bytes(self, pk_bytes, out=None):
        """PublicKey::try_from_bytes for a batch -> PublicKeys (expand_public, src/ml_dsa.rs:477).
        out: a PublicKeys from empty_public_keys() to fill (same buffers every call: a repeated call shape)."""
        pk = self._key_bytes(pk_bytes, self.PK_LEN, "pk")
        n = pk.shape[0]
        o = out or self.empty_public_keys(n)
        _lib.check(self.lib.mldsa_pk_expand(self.hp._h, self.pset, _ptr(pk), _ptr(o.rho), _ptr(o.tr), _ptr(o.t1_d2_hat_mont), n, _stream(self.device)))
        return o

    def private_keys_from_bytes(self, sk_bytes, out=None):
        """PrivateKey::try_from_bytes for a batch -> PrivateKeys (expand_private, src/ml_dsa.rs:445)"""
        sk = self._key_bytes(sk_bytes, self.SK_LEN, "sk")
        n = sk.shape[0]
        o = out or self.empty_private_keys(n)
        _lib.check(self.lib.mldsa_sk_expand(self.hp._h, self.pset, _ptr(sk), _ptr(o.rho), _ptr(o.cap_k), _ptr(o.tr), _ptr(o.s_1_hat_mont),
                                            _ptr(o.s_2_hat_mont), _ptr(o.t_0_hat_mont), n, _stream(self.device)))
        return o

    def public_keys_into_bytes(self, pks):
        """PublicKey::into_bytes for a batch (src/lib.rs:478-493): uint8 tensor [n, PK_LEN]"""
        n = len(pks)
        pk = torch.empty((n, self.PK_LEN), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.mldsa_pk_into_bytes(self.hp._h, self.pset, _ptr(pks.rho), _ptr(pks.t1_d2_hat_mont), _ptr(pk), n, _stream(self.device)))
        return pk

    def private_keys_into_bytes(self, sks):
        """PrivateKey::into_bytes for a batch (src/lib.rs:427-465): uint8 tensor [n, SK_LEN]"""
        n = len(sks)
        sk = torch.empty((n, self.SK_LEN), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.mldsa_sk_into_bytes(self.hp._h, self.pset, _ptr(sks.rho), _ptr(sks.cap_k), _ptr(sks.tr),
                                                _ptr(sks.s_1_hat_mont), _ptr(sks.s_2_hat_mont), _ptr(sks.t_0_hat_mont), _ptr(sk), n,
                                                _stream(self.device)))
        return sk

    def get_public_key(self, sks):
        """PrivateKey::get_public_key for a batch (src/lib.rs:345-349 -> private_to_public_key, ml_dsa.rs:502-559)"""
        n, k = len(sks), self.params.k
        rho = torch.empty((n, 32), dtype=torch.uint8, device=self.device)
        tr = torch.empty((n, 64), dtype=torch.uint8, device=self.device)
        t1 = torch.empty((n, k, N), dtype=torch.int32, device=self.device)
        _lib.check(self.lib.mldsa_get_public_key(self.hp._h, self.pset, _ptr(sks.rho), _ptr(sks.tr), _ptr(sks.s_1_hat_mont),
                                                 _ptr(sks.s_2_hat_mont), _ptr(rho), _ptr(tr), _ptr(t1), n, _stream(self.device)))
        return PublicKeys(self.pset, rho, tr, t1)

    # ---- host-memory entry points (numpy arrays in, numpy arrays out; staging inside the library) --------
    def _host_fn(self, op):
        return getattr(self.lib, f"mldsa_{op}_host")

    def _host_handle(self):
        return self.hp._h

    @staticmethod
    def _np_u8(a, row, what):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        if row and a.size % row:
            raise ValueError(f"{what}: length is not a multiple of {row}")
        return a

    @staticmethod
    def _cat_host(items):
        lens = np.fromiter((len(b) for b in items), dtype=np.uint64, count=len(items))
        off = np.zeros(len(items) + 1, dtype=np.uint64)
        np.cumsum(lens, out=off[1:])
        flat = np.frombuffer(b"".join(bytes(b) for b in items) or b"\0", dtype=np.uint8)
        return flat, off

    @staticmethod
    def _host_strings(items, n_ops, what):
        """list of byte strings or (flat uint8, uint64 offsets[n_ops + 1]) -> validated contiguous (flat, offsets).
        The library walks offsets[0 .. n_ops] and reads flat[offsets[0] .. offsets[n_ops]): both are checked here, so a
        short list or a truncated buffer is a ValueError and never an out-of-bounds read on the C side."""
        if isinstance(items, tuple):
            flat, off = items
            flat = np.ascontiguousarray(flat, dtype=np.uint8)
            off = np.ascontiguousarray(off)
            if off.dtype != np.uint64:
                if off.size and (not np.issubdtype(off.dtype, np.integer) or int(off.min()) < 0):
                    raise ValueError(f"{what}: offsets must be non-negative integers")
                off = off.astype(np.uint64)
        else:
            if len(items) != n_ops:
                raise ValueError(f"{what}: {len(items)} entries for {n_ops} operations")
            flat, off = MlDsa._cat_host(items)
        if off.ndim != 1 or off.size != n_ops + 1:
            raise ValueError(f"{what}: offsets must have n_ops + 1 = {n_ops + 1} entries, got {off.size}")
        if n_ops and bool(np.any(off[1:] < off[:-1])):
            raise ValueError(f"{what}: offsets must be non-decreasing")
        if int(off[-1]) > flat.size:
            raise ValueError(f"{what}: offsets run past the end of the byte buffer ({int(off[-1])} > {flat.size})")
        return flat, off

    @staticmethod
    def _host_out(a, dtype, n_items, what):
        if not isinstance(a, np.ndarray) or a.dtype != dtype or not a.flags.c_contiguous or not a.flags.writeable or a.size < n_items:
            raise ValueError(f"{what}: a writable C-contiguous {np.dtype(dtype).name} array of at least {n_items} elements is required")
        return a

    def verify_host(self, pk_bytes, messages, sigs, ctxs=None, key_idx=None, mode=MODE_PURE, out=None):
        """mldsa_verify_host: wire-format public keys [n_keys, PK_LEN], signatures [n_ops, SIG_LEN] and messages
        in HOST memory (numpy); returns a bool array.  `messages` / `ctxs`: list of byte strings, or a
        (flat uint8 array, uint64 offsets[n + 1]) pair."""
        pk = self._np_u8(pk_bytes, self.PK_LEN, "pk")
        sg = self._np_u8(sigs, self.SIG_LEN, "sigs")
        n_keys, n_ops = pk.size // self.PK_LEN, sg.size // self.SIG_LEN
        mflat, moff = self._host_strings(messages, n_ops, "messages")
        cflat = coff = None
        if ctxs is not None:
            cflat, coff = self._host_strings(ctxs, n_ops, "ctxs")
        kidx = _check_key_idx(key_idx, n_keys, n_ops)
        # out: caller's (page-locked) uint8[n_ops]
        ok = self._host_out(out, np.uint8, n_ops, "verify_host: out") if out is not None else np.zeros(max(n_ops, 1), dtype=np.uint8)
        vp = lambda a: C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)
        _lib.check(self._host_fn("verify")(self._host_handle(), self.pset, mode, vp(pk), n_keys, vp(kidx), vp(mflat), vp(moff), vp(cflat),
                                           vp(coff), vp(sg), vp(ok), n_ops))
        return ok[:n_ops].astype(bool)

    def sign_host(self, sk_bytes, messages, rnd, ctxs=None, key_idx=None, mode=MODE_PURE, out=None):
        """mldsa_sign_host: wire-format private keys, messages and rnd in HOST memory; returns uint8 [n_ops, SIG_LEN].
        out: (sig uint8[n_ops, SIG_LEN], status int32[n_ops]) buffers of the caller (page-locked ones are filled by DMA)."""
        sk = self._np_u8(sk_bytes, self.SK_LEN, "sk")
        rn = self._np_u8(rnd, 32, "rnd")
        n_keys, n_ops = sk.size // self.SK_LEN, rn.size // 32
        mflat, moff = self._host_strings(messages, n_ops, "messages")
        cflat = coff = None
        if ctxs is not None:
            cflat, coff = self._host_strings(ctxs, n_ops, "ctxs")
        kidx = _check_key_idx(key_idx, n_keys, n_ops)
        if out is not None:
            if not isinstance(out, tuple) or len(out) != 2:
                raise ValueError("sign_host: out = (sig uint8[n_ops, SIG_LEN], status int32[n_ops])")
            sig = self._host_out(out[0], np.uint8, n_ops * self.SIG_LEN, "sign_host: out[0] (signatures)")
            status = self._host_out(out[1], np.int32, n_ops, "sign_host: out[1] (status)")
            if sig.ndim == 2 and sig.shape[1] != self.SIG_LEN:
                raise ValueError(f"sign_host: out[0] rows must be SIG_LEN = {self.SIG_LEN} bytes")
            sig = sig.reshape(-1)[:n_ops * self.SIG_LEN].reshape(n_ops, self.SIG_LEN) if n_ops else sig
        else:
            sig, status = np.zeros((max(n_ops, 1), self.SIG_LEN), dtype=np.uint8), np.zeros(max(n_ops, 1), dtype=np.int32)
        vp = lambda a: C.c_void_p(a.ctypes.data) if a is not None else C.c_void_p(0)
        _lib.check(self._host_fn("sign")(self._host_handle(), self.pset, mode, vp(sk), n_keys, vp(kidx), vp(mflat), vp(moff), vp(cflat),
                                         vp(coff), vp(rn), vp(sig), vp(status), n_ops))
        if n_ops and int(status[:n_ops].min()) < 0:
            raise ValueError("ML-DSA.Sign: ctx too long")
        return sig[:n_ops]

    def keygen_host(self, xi, out=None):
        """mldsa_keygen_host: seeds [n, 32] in host memory -> (pk [n, PK_LEN], sk [n, SK_LEN]) numpy arrays.
        out: (pk, sk) uint8 arrays to fill (page-locked ones are written by the DMA directly)."""
        x = self._np_u8(xi, 32, "xi")
        n = x.size // 32
        if out is not None:
            pk, sk = out
            if pk.dtype != np.uint8 or sk.dtype != np.uint8 or pk.size < n * self.PK_LEN or sk.size < n * self.SK_LEN \
                    or not pk.flags.c_contiguous or not sk.flags.c_contiguous:
                raise ValueError("keygen_host: out = (pk, sk) contiguous uint8 arrays of at least n keys")
        else:
            pk = np.zeros((max(n, 1), self.PK_LEN), dtype=np.uint8)
            sk = np.zeros((max(n, 1), self.SK_LEN), dtype=np.uint8)
        vp = lambda a: C.c_void_p(a.ctypes.data)
        _lib.check(self._host_fn("keygen")(self._host_handle(), self.pset, vp(x), vp(pk), vp(sk), n))
        return pk[:n], sk[:n]

    # ---- KeyGen (src/traits.rs:8-114; src/lib.rs:247-250) ------------------------------
    def keygen_from_seed(self, xi, out=None):
        """KG::keygen_from_seed for a batch of 32-byte seeds -> (pk bytes, sk bytes) tensors
        [n, PK_LEN] / [n, SK_LEN] in FIPS 204 wire format (= the reference's into_bytes()).
        out: (pk, sk) tensors to fill."""
        xi = self._key_bytes(xi, 32, "xi")
        n = xi.shape[0]
        pk, sk = out or (torch.empty((n, self.PK_LEN), dtype=torch.uint8, device=self.device),
                         torch.empty((n, self.SK_LEN), dtype=torch.uint8, device=self.device))
        _lib.check(self.lib.mldsa_keygen(self.hp._h, self.pset, _ptr(xi), _ptr(pk), _ptr(sk), n, _stream(self.device)))
        return pk, sk

    def try_keygen_with_rng(self, rng, n=1):
        """KG::try_keygen_with_rng: xi drawn from the caller's rng (rng.fill_bytes(32) per key)"""
        xi = [rng.fill_bytes(32) for _ in range(n)]
        return self.keygen_from_seed(xi)

    def try_keygen(self, n=1):
        """KG::try_keygen (src/traits.rs:44-46): xi from the operating system's generator (OsRng)"""
        return self.try_keygen_with_rng(OsRng(), n)

    # ---- Signer (src/traits.rs:118-308; src/lib.rs:268-342, 586-600) -------------------
    def try_sign_with_rng(self, rng, sks, messages, ctxs=None, key_idx=None):
        """PrivateKey::try_sign_with_rng (src/lib.rs:268-296): one rnd = rng.fill_bytes(32) per op, drawn only after
        every ctx passed the length check (lib.rs:274 comes before lib.rs:282)"""
        if ctxs is not None and any(len(c) > 255 for c in ctxs):
            raise ValueError("ML-DSA.Sign: ctx too long")
        return self.try_sign_with_seed(sks, messages, [rng.fill_bytes(32) for _ in messages], ctxs=ctxs, key_idx=key_idx)

    def try_sign(self, sks, messages, ctxs=None, key_idx=None):
        """PrivateKey::try_sign (src/traits.rs:156-158): hedged signing with rnd from OsRng"""
        return self.try_sign_with_rng(OsRng(), sks, messages, ctxs=ctxs, key_idx=key_idx)

    def try_hash_sign_with_rng(self, rng, sks, messages, ctxs=None, ph=PH_SHA512, key_idx=None):
        """PrivateKey::try_hash_sign_with_rng (src/lib.rs:310-342)"""
        if ctxs is not None and any(len(c) > 255 for c in ctxs):
            raise ValueError("HashML-DSA.Sign: ctx too long")
        return self.try_hash_sign_with_seed(sks, messages, [rng.fill_bytes(32) for _ in messages], ctxs=ctxs, ph=ph, key_idx=key_idx)

    def try_hash_sign(self, sks, messages, ctxs=None, ph=PH_SHA512, key_idx=None):
        """PrivateKey::try_hash_sign (src/traits.rs:247-251)"""
        return self.try_hash_sign_with_rng(OsRng(), sks, messages, ctxs=ctxs, ph=ph, key_idx=key_idx)

    def try_sign_with_seed(self, sks, messages, rnd, ctxs=None, key_idx=None, mode=MODE_PURE):
        """PrivateKey::try_sign_with_seed for a batch: rnd = one 32-byte seed per op (zeros =
        deterministic signing).  Raises ValueError if any ctx is longer than 255 bytes
        (src/lib.rs:274).  Returns a uint8 tensor [n_ops, SIG_LEN]."""
        n_ops = len(messages)
        msg_buf, msg_off = _cat_with_offsets(messages, self.device)
        ctx_buf = ctx_off = None
        if ctxs is not None:
            ctx_buf, ctx_off = _cat_with_offsets(ctxs, self.device)
        if key_idx is None and len(sks) != n_ops:
            key_idx = np.arange(n_ops, dtype=np.uint32) % len(sks)
        key_idx = _check_key_idx(key_idx, len(sks), n_ops)
        kidx = None
        if key_idx is not None:
            kidx = torch.as_tensor(key_idx.view(np.int32)).to(self.device)
        rnd = self._key_bytes(rnd, 32, "rnd") if n_ops else torch.zeros((1, 32), dtype=torch.uint8, device=self.device)
        sigs = torch.empty((max(n_ops, 1), self.SIG_LEN), dtype=torch.uint8, device=self.device)
        status = torch.zeros(max(n_ops, 1), dtype=torch.int32, device=self.device)
        self.sign_device(sks, msg_buf, msg_off, rnd, sigs, n_ops, ctx_buf, ctx_off, kidx, mode, status)
        torch.cuda.synchronize(self.device)
        if n_ops and int(status[:n_ops].min()) < 0:
            st = status[:n_ops].cpu().numpy()
            bad = int(np.flatnonzero(st < 0)[0])
            code = int(st[bad])
            what = {_lib.ERR_CTX_LEN: "ML-DSA.Sign: ctx too long",
                    _lib.ERR_PARAM: "ML-DSA.Sign: operation refused (malformed offsets or key index out of range)",
                    _lib.ERR_AGAIN: "ML-DSA.Sign: operation left unfinished by an asynchronous call"}.get(code, f"ML-DSA.Sign: status {code}")
            raise ValueError(f"{what} (op {bad})")
        return sigs[:n_ops]

    def try_hash_sign_with_seed(self, sks, messages, rnd, ctxs=None, ph=PH_SHA512, key_idx=None):
        """PrivateKey::try_hash_sign_with_seed (src/traits.rs:280-284, src/lib.rs:310-342) for a batch: HashML-DSA.Sign
        with the pre-hash `ph` computed on the host."""
        return self.try_sign_with_seed(sks, [hash_message(m, ph) for m in messages], rnd, ctxs=ctxs, key_idx=key_idx,
                                       mode=MODE_PREHASH)

    def sign_device(self, sks, msg_buf, msg_off, rnd, sigs, n_ops, ctx_buf=None, ctx_off=None, key_idx=None,
                    mode=MODE_PURE, status=None, a_hat=None, wait=True):
        """Everything already resident in HBM (what bench.py times).  wait=False -> mldsa_sign_async: the call
        only enqueues; an op the enqueued rounds leave unfinished (p < 1e-9 per call) has status
        MLDSA_ERR_AGAIN and must be signed again."""
        null = C.c_void_p(0)
        if a_hat is not None:
            fn, first = self.lib.mldsa_sign_cached_a, a_hat
        else:
            fn, first = (self.lib.mldsa_sign if wait else self.lib.mldsa_sign_async), sks.rho
        _lib.check(fn(
            self.hp._h, self.pset, mode, _ptr(first), _ptr(sks.cap_k), _ptr(sks.tr), _ptr(sks.s_1_hat_mont),
            _ptr(sks.s_2_hat_mont), _ptr(sks.t_0_hat_mont), len(sks), _ptr(key_idx) if key_idx is not None else null,
            _ptr(msg_buf), _ptr(msg_off), _ptr(ctx_buf) if ctx_buf is not None else null,
            _ptr(ctx_off) if ctx_off is not None else null, _ptr(rnd), _ptr(sigs),
            _ptr(status) if status is not None else null, n_ops, _stream(self.device)))
        return sigs


class MlDsaGroup(MlDsa):
    """The host-memory entry points over SEVERAL GPUs of one node: mldsa_group_create(device_ids) + mldsa_*_host_group
    (include/mldsa_hip.h "several GPUs of one node").  verify_host / sign_host / keygen_host take exactly the arguments of
    MlDsa's and return byte-identical results: the library cuts the batch into contiguous slices of ceil(B / N) ops, one
    worker thread and context per entry of `device_ids` (a device may be listed twice: two contexts on one GPU), results
    land directly in the caller's arrays.  Needs no torch: every pointer is host memory."""

    def __init__(self, pset, device_ids):
        self.pset = pset
        self.lib = _lib.load()
        p = _lib.get_params(pset)
        self.params = p
        self.PK_LEN, self.SK_LEN, self.SIG_LEN = p.pk_len, p.sk_len, p.sig_len
        self.device_ids = [int(d) for d in device_ids]
        ids = (C.c_int * len(self.device_ids))(*self.device_ids)
        g = C.c_void_p()
        _lib.check(self.lib.mldsa_group_create(ids, len(self.device_ids), C.byref(g)))
        self._g = g
        self.hp = None

    def __len__(self):
        return len(self.device_ids)

    def close(self):
        if getattr(self, "_g", None):
            self.lib.mldsa_group_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _host_fn(self, op):
        return getattr(self.lib, f"mldsa_{op}_host_group")

    def _host_handle(self):
        return self._g

    def ctx(self, i):
        """the i-th context of the group as a raw handle (mldsa_set_option / mldsa_reserve)"""
        return C.c_void_p(self.lib.mldsa_group_ctx(self._g, i))

    def set_option(self, option, value):
        for i in range(len(self)):
            _lib.check(self.lib.mldsa_set_option(self.ctx(i), option, value))

    # ---- device-resident slices: one process, one thread, N devices (mldsa_*_group) --------------------------
    def on_device(self, i):
        """an MlDsa bound to the i-th context of the group (its device, its workspace): expand keys and stage a slice's
        inputs there, then hand the slices to verify_group / sign_group / keygen_group"""
        h = HotPath.from_handle(self.ctx(i), self.device_ids[i])
        return MlDsa(self.pset, hotpath=h)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)

    @staticmethod
    def _slice_stream(t, stream):
        if stream is not None:
            return C.c_void_p(stream)
        with torch.cuda.device(t.device):
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def verify_group(self, slices, mode=MODE_PURE, wait=True):
        """mldsa_verify_group.  slices[i]: dict(pks=PublicKeys on device i, msg_buf, msg_off, sigs, ok, n_ops and optionally
        ctx_buf, ctx_off, key_idx, stream) -- the arguments of MlDsa.verify_device for slice i, tensors on device i."""
        arr = (_lib.VerifySlice * len(self))()
        for i, sl in enumerate(slices):
            pks = sl["pks"]
            arr[i] = _lib.VerifySlice(self._p(pks.rho), self._p(pks.tr), self._p(pks.t1_d2_hat_mont), len(pks), self._p(sl.get("key_idx")),
                                      self._p(sl["msg_buf"]), self._p(sl["msg_off"]), self._p(sl.get("ctx_buf")), self._p(sl.get("ctx_off")),
                                      self._p(sl["sigs"]), self._p(sl["ok"]), sl["n_ops"], self._slice_stream(sl["ok"], sl.get("stream")))
        _lib.check(self.lib.mldsa_verify_group(self._g, self.pset, mode, arr, 1 if wait else 0))

    def sign_group(self, slices, mode=MODE_PURE, wait=True):
        """mldsa_sign_group.  slices[i]: dict(sks=PrivateKeys on device i, msg_buf, msg_off, rnd, sigs, status, n_ops and
        optionally ctx_buf, ctx_off, key_idx, stream); wait=False signs with mldsa_sign_async semantics."""
        arr = (_lib.SignSlice * len(self))()
        for i, sl in enumerate(slices):
            sks = sl["sks"]
            arr[i] = _lib.SignSlice(self._p(sks.rho), self._p(sks.cap_k), self._p(sks.tr), self._p(sks.s_1_hat_mont), self._p(sks.s_2_hat_mont),
                                    self._p(sks.t_0_hat_mont), len(sks), self._p(sl.get("key_idx")), self._p(sl["msg_buf"]), self._p(sl["msg_off"]),
                                    self._p(sl.get("ctx_buf")), self._p(sl.get("ctx_off")), self._p(sl["rnd"]), self._p(sl["sigs"]),
                                    self._p(sl.get("status")), sl["n_ops"], self._slice_stream(sl["sigs"], sl.get("stream")))
        _lib.check(self.lib.mldsa_sign_group(self._g, self.pset, mode, arr, 1 if wait else 0))

    def keygen_group(self, slices, wait=True):
        """mldsa_keygen_group.  slices[i]: dict(xi, pk, sk, n_keys[, stream]), tensors on device i."""
        arr = (_lib.KeygenSlice * len(self))()
        for i, sl in enumerate(slices):
            arr[i] = _lib.KeygenSlice(self._p(sl["xi"]), self._p(sl["pk"]), self._p(sl["sk"]), sl["n_keys"], self._slice_stream(sl["pk"], sl.get("stream")))
        _lib.check(self.lib.mldsa_keygen_group(self._g, self.pset, arr, 1 if wait else 0))

    def sync(self):
        """mldsa_group_sync: waits for the streams of the last device-resident group call"""
        _lib.check(self.lib.mldsa_group_sync(self._g))

    def allgather(self, bufs, n_ops, use_rccl=-1):
        """mldsa_group_allgather over one uint8 tensor per device (N * ceil(n_ops / N) bytes each, slice i of bufs[i] filled)"""
        arr = (C.c_void_p * len(self))(*[b.data_ptr() for b in bufs])
        _lib.check(self.lib.mldsa_group_allgather(self._g, arr, n_ops, use_rccl))

    def shard(self, n_ops, part):
        """(first, count) of the slice part `part` owns: mldsa_group_shard (= multi_gpu.shard)"""
        a, c = C.c_size_t(), C.c_size_t()
        _lib.check(self.lib.mldsa_group_shard(n_ops, len(self), part, C.byref(a), C.byref(c)))
        return a.value, c.value


class MlDsaBatcher:
    """mldsa_batcher_*: the reference's ONE-operation-per-call surface (src/traits.rs:118-308, 330-362) for many host threads at once.
    verify / sign / keygen block the calling thread (ctypes releases the GIL for the duration), the library coalesces whatever
    the threads submit into batched calls on one context.  Byte strings in, byte strings / bool out."""

    def __init__(self, pset, device=0, max_batch=4096, max_wait_us=0, cache_keys=0, hotpath=None, device_ids=None):
        """device_ids: mldsa_batcher_create_on -- one lane (context + dispatcher thread + key table) per entry, owned by the batcher;
        otherwise one lane on `hotpath`'s context (or a new one on `device`)."""
        self.pset = pset
        p = _lib.get_params(pset)
        self.PK_LEN, self.SK_LEN, self.SIG_LEN = p.pk_len, p.sk_len, p.sig_len
        h = C.c_void_p()
        if device_ids is not None:
            self.hp = None
            self.lib = _lib.load()
            ids = (C.c_int * len(device_ids))(*device_ids)
            _lib.check(self.lib.mldsa_batcher_create_on(ids, len(device_ids), pset, max_batch, max_wait_us, cache_keys, C.byref(h)))
        else:
            self.hp = hotpath or HotPath(device)
            self.lib = self.hp.lib
            _lib.check(self.lib.mldsa_batcher_create(self.hp._h, pset, max_batch, max_wait_us, cache_keys, C.byref(h)))
        self._b = h

    def close(self):
        if getattr(self, "_b", None):
            self.lib.mldsa_batcher_destroy(self._b)
            self._b = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _fixed(b, n, what):
        b = bytes(b)
        if len(b) != n:
            raise ValueError(f"{what}: {len(b)} bytes, expected {n}")
        return b

    def verify(self, pk, message, sig, ctx=b"", mode=MODE_PURE):
        """PublicKey::try_from_bytes(pk)?.verify(message, sig, ctx) (src/lib.rs:364-380)"""
        pk, sig = self._fixed(pk, self.PK_LEN, "pk"), self._fixed(sig, self.SIG_LEN, "sig")
        message, ctx = bytes(message), bytes(ctx)
        ok = C.c_uint8(0)
        _lib.check(self.lib.mldsa_batcher_verify(self._b, mode, pk, message, len(message), ctx, len(ctx), sig, C.byref(ok)))
        return bool(ok.value)

    def sign(self, sk, message, rnd, ctx=b"", mode=MODE_PURE):
        """PrivateKey::try_from_bytes(sk)?.try_sign_with_seed(rnd, message, ctx) (src/lib.rs:268-296); raises MldsaError for |ctx| > 255"""
        sk, rnd = self._fixed(sk, self.SK_LEN, "sk"), self._fixed(rnd, 32, "rnd")
        message, ctx = bytes(message), bytes(ctx)
        sig = (C.c_uint8 * self.SIG_LEN)()
        _lib.check(self.lib.mldsa_batcher_sign(self._b, mode, sk, message, len(message), ctx, len(ctx), rnd, sig))
        return bytes(sig)

    def keygen_from_seed(self, xi):
        """KG::keygen_from_seed(xi) (src/lib.rs:247-250) -> (pk bytes, sk bytes)"""
        xi = self._fixed(xi, 32, "xi")
        pk, sk = (C.c_uint8 * self.PK_LEN)(), (C.c_uint8 * self.SK_LEN)()
        _lib.check(self.lib.mldsa_batcher_keygen(self._b, xi, pk, sk))
        return bytes(pk), bytes(sk)

    def forget_key(self, key):
        """take one key (pk or sk wire bytes) out of the device-resident tables: the host copy is cleared, a private key's device fields zeroed"""
        key = bytes(key)
        _lib.check(self.lib.mldsa_batcher_forget_key(self._b, key, len(key)))

    def flush_keys(self):
        _lib.check(self.lib.mldsa_batcher_flush_keys(self._b))

    def set_private_key_cache(self, on):
        """False: no private key stays in the table beyond the batch that used it (the caller's zeroize then ends the key's life)"""
        _lib.check(self.lib.mldsa_batcher_set_private_key_cache(self._b, 1 if on else 0))

    def stats(self):
        st = _lib.BatcherStats()
        _lib.check(self.lib.mldsa_batcher_get_stats(self._b, C.byref(st)))
        return {n: int(getattr(st, n)) for n, _ in st._fields_}
