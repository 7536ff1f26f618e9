"""The wire-format codecs as seams of their own: mldsa_bit_pack / bit_unpack / hint_bit_pack /
hint_bit_unpack / sig_encode / sig_decode / w1_encode against the oracle's restatements of src/conversion.rs and src/encodings.rs,
on the cases the reference's own unit tests use (conversion.rs:490-640: round trips of random bytes, zero polynomials, parameter
ranges) and on every way a hint section can be malformed (conversion.rs:364-399)."""
import ctypes as C

import numpy as np
import pytest
import torch

from fips204_amd import _lib
from fips204_amd.hotpath import HotPath
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
N = 256


@pytest.fixture(scope="module")
def hp():
    return HotPath(0)


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def dev(a):
    return torch.from_numpy(np.array(a, copy=True)).cuda()


# (a, b) pairs the reference packs with: t1 (encodings.rs:38), eta (137-150), t0 (155-160), z (266), w1 (354), and the 6-bit
# round trip of conversion.rs:491-499
PAIRS = [(0, 1023), (2, 2), (4, 4), (4095, 4096), ((1 << 17) - 1, 1 << 17), ((1 << 19) - 1, 1 << 19), (0, 15), (0, 43), (0, 63)]


@pytest.mark.parametrize("a,b", PAIRS)
def test_bit_pack_unpack(hp, a, b):
    """bit_pack / bit_unpack (conversion.rs:143-262) and their a = 0 forms simple_bit_pack / simple_bit_unpack: random
    polynomials in [-a, b] byte-exact with the oracle, random BYTES unpacked like the oracle (verdict included: (0, 43) and (2, 2),
    (4, 4) are the pairs whose fields can leave [-a, b]), and the reference's round-trip test."""
    rng = np.random.default_rng(a * 31 + b)
    bitlen = int(a + b).bit_length()
    n = 96
    w = rng.integers(-a, b + 1, (n, N)).astype(np.int32)
    w[0] = 0                      # conversion.rs:524-530
    w[1] = b
    w[2] = -a
    got = host(hp.bit_pack(dev(w), a, b))
    assert got.shape == (n, 32 * bitlen)
    for i in range(n):
        assert got[i].tobytes() == orc.bit_pack(w[i], a, b, 32 * bitlen), i
    back, ok = hp.bit_unpack(dev(got), a, b)
    assert np.array_equal(host(back), w) and host(ok).all()
    raw = rng.integers(0, 256, (n, 32 * bitlen), dtype=np.uint8)
    raw[0] = 0xFF
    raw[1] = 0
    uw, uok = hp.bit_unpack(dev(raw), a, b)
    uw, uok = host(uw), host(uok)
    n_err = 0
    for i in range(n):
        o_ok, o_w = orc.bit_unpack(raw[i].tobytes(), a, b)
        assert bool(uok[i]) == o_ok and np.array_equal(uw[i], o_w), i
        n_err += not o_ok
    if a == 0 and (b + 1) & b:        # b + 1 not a power of two: fields above b exist (the Err of conversion.rs:260)
        assert n_err > 0
    if (a + b + 1) & (a + b) == 0:    # every field value is a coefficient: bytes -> w -> bytes is the identity (conversion.rs:491-499)
        assert np.array_equal(host(hp.bit_pack(dev(uw), a, b)), raw)


def test_bit_pack_argument_ranges(hp):
    """conversion.rs:532-575: `a`, `b` outside the reference's debug_asserts are MLDSA_ERR_PARAM, not a launch"""
    w = dev(np.zeros((1, N), dtype=np.int32))
    out = torch.zeros(32 * 21, dtype=torch.uint8, device="cuda")
    p, q = C.c_void_p(w.data_ptr()), C.c_void_p(out.data_ptr())
    for a, b in ((0, 0), (-1, 5), (1 << 20, 1), (1, 1 << 20)):
        assert hp.lib.mldsa_bit_pack(hp._h, p, a, b, q, 1, None) == _lib.ERR_PARAM
        assert hp.lib.mldsa_bit_unpack(hp._h, q, a, b, p, None, 1, None) == _lib.ERR_PARAM
    assert hp.lib.mldsa_bit_pack(hp._h, None, 0, 1, None, 0, None) == _lib.OK
    assert hp.lib.mldsa_bit_pack(hp._h, None, 0, 1, q, 1, None) == _lib.ERR_PARAM


def random_hints(rng, k, omega, n, full=False):
    h = np.zeros((n, k, N), dtype=np.int32)
    for i in range(n):
        weight = omega if full or i % 5 == 0 else int(rng.integers(0, omega + 1))
        pos = rng.choice(k * N, weight, replace=False)
        h[i].reshape(-1)[pos] = 1
    h[1] = 0
    return h


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_hint_bit_pack_unpack(hp, pset):
    """hint_bit_pack / hint_bit_unpack (conversion.rs:277-414): weights 0 .. omega (all in one polynomial, in the last one, spread),
    byte-exact with the oracle and back; weight omega + 1 is refused (ok = 0) without writing past the position bytes."""
    p = orc.params(pset)
    k, omega = p.k, p.omega
    rng = np.random.default_rng(pset)
    h = random_hints(rng, k, omega, 64)
    h[2] = 0
    h[2, 0, :omega] = 1
    h[3] = 0
    h[3, k - 1, N - omega:] = 1
    h[4] = 0
    h[4, :, 255] = 1
    y, ok = hp.hint_bit_pack(pset, dev(h))
    y, ok = host(y), host(ok)
    assert ok.all()
    lib = orc.lib()
    for i in range(h.shape[0]):
        want = (C.c_uint8 * (omega + k))()
        lib.orc_hint_bit_pack(k, omega, np.ascontiguousarray(h[i]).ctypes.data_as(C.c_void_p), want)
        assert y[i].tobytes() == bytes(want), i
    back, bok = hp.hint_bit_unpack(pset, dev(y))
    assert host(bok).all() and np.array_equal(host(back), h)
    over = np.zeros((2, k, N), dtype=np.int32)
    over[0, 0, :omega + 1] = 1
    over[1, :, :(omega // k) + 1] = 1
    y2, ok2 = hp.hint_bit_pack(pset, dev(over))
    y2 = host(y2)
    assert not host(ok2).any() and (y2[:, omega:] <= omega).all()


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_hint_bit_unpack_malformed(hp, pset):
    """Every Err of hint_bit_unpack (conversion.rs:364-399) and its neighbours that are Ok, verdict AND polynomials against the
    oracle: a limit below its predecessor, a limit above omega, positions equal / decreasing inside a polynomial (but not across
    two), non-zero padding; then 4096 random mutations of valid sections."""
    p = orc.params(pset)
    k, omega = p.k, p.omega
    rng = np.random.default_rng(1000 + pset)
    base_h = random_hints(rng, k, omega, 8)
    base, _ = hp.hint_bit_pack(pset, dev(base_h))
    base = host(base)
    cases = []
    y = np.zeros(omega + k, dtype=np.uint8); y[0] = 5; y[omega] = 1; y[omega + 2:] = 1; cases.append(y)   # limit 1 (= 0) below limit 0
    y = base[2].copy(); y[omega + k - 1] = omega + 1; cases.append(y)
    y = base[2].copy(); y[omega + k - 1] = 255; cases.append(y)
    y = np.zeros(omega + k, dtype=np.uint8); y[0:2] = (7, 7); y[omega:] = 2; cases.append(y)          # equal positions
    y = np.zeros(omega + k, dtype=np.uint8); y[0:2] = (9, 7); y[omega:] = 2; cases.append(y)          # decreasing
    y = np.zeros(omega + k, dtype=np.uint8); y[0:2] = (9, 7); y[omega] = 1; y[omega + 1:] = 2; cases.append(y)  # across two polynomials: Ok
    y = np.zeros(omega + k, dtype=np.uint8); y[omega - 1] = 1; cases.append(y)                        # padding
    y = np.zeros(omega + k, dtype=np.uint8); y[0] = 0; y[1] = 0; y[omega:] = 1; y[1] = 3; cases.append(y)  # padding right behind the last position
    y = np.zeros(omega + k, dtype=np.uint8); y[:omega] = np.arange(omega); y[omega:] = omega; cases.append(y)   # full, Ok
    y = np.zeros(omega + k, dtype=np.uint8); cases.append(y)
    mut = base[rng.integers(0, 8, 4096)].copy()
    pos = rng.integers(0, omega + k, 4096)
    mut[np.arange(4096), pos] = rng.integers(0, 256, 4096)
    ys = np.concatenate([np.stack(cases), mut])
    h, ok = hp.hint_bit_unpack(pset, dev(ys))
    h, ok = host(h), host(ok)
    verdicts = []
    for i in range(ys.shape[0]):
        o_ok, o_h = orc.hint_bit_unpack(k, omega, ys[i].tobytes())
        verdicts.append(o_ok)
        assert bool(ok[i]) == o_ok, (i, ys[i, omega:])
        assert np.array_equal(h[i], o_h if o_ok else np.zeros_like(o_h)), i
    v = np.array(verdicts)
    assert list(v[:10]) == [False, False, False, False, False, True, False, False, True, True]
    assert 0 < v[10:].sum() < 4096


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_sig_encode_decode(hp, pset):
    """sig_encode / sig_decode (encodings.rs:238-328): random (c_tilde, z, h) with z over the whole of (-gamma1, gamma1], byte-exact
    with the oracle; real signatures from the signer decode to what the oracle decodes and re-encode to themselves; damaged hint
    sections give the oracle's verdict; z = -gamma1 (outside the encoder's contract) is flagged."""
    p = orc.params(pset)
    rng = np.random.default_rng(77 + pset)
    n = 48
    ct = rng.integers(0, 256, (n, p.ctilde_len), dtype=np.uint8)
    z = rng.integers(-p.gamma1 + 1, p.gamma1 + 1, (n, p.l, N)).astype(np.int32)
    z[0] = p.gamma1
    z[1] = -p.gamma1 + 1
    z[2] = 0
    h = random_hints(rng, p.k, p.omega, n)
    sigs, ok = hp.sig_encode(pset, dev(ct), dev(z), dev(h))
    sigs = host(sigs)
    assert host(ok).all() and sigs.shape == (n, p.sig_len)
    lib = orc.lib()
    for i in range(n):
        want = (C.c_uint8 * p.sig_len)()
        lib.orc_sig_encode(pset, orc._u8(ct[i].tobytes()), np.ascontiguousarray(z[i]).ctypes.data_as(C.c_void_p),
                           np.ascontiguousarray(h[i]).ctypes.data_as(C.c_void_p), want)
        assert sigs[i].tobytes() == bytes(want), i
    dct, dz, dh, dok = (host(t) for t in hp.sig_decode(pset, dev(sigs)))
    assert dok.all() and np.array_equal(dct, ct) and np.array_equal(dz, z) and np.array_equal(dh, h)
    # arbitrary bytes: z always decodes (2 gamma1 is a power of two), the hint section rarely
    raw = rng.integers(0, 256, (256, p.sig_len), dtype=np.uint8)
    raw[:128, -(p.omega + p.k):] = sigs[rng.integers(0, n, 128), -(p.omega + p.k):]
    raw[64:128, rng.integers(p.sig_len - p.omega - p.k, p.sig_len, 64)] ^= 1
    dct, dz, dh, dok = (host(t) for t in hp.sig_decode(pset, dev(raw)))
    n_ok = 0
    for i in range(raw.shape[0]):
        o_ok, o_ct, o_z, o_h = orc.sig_decode(pset, raw[i].tobytes())
        assert bool(dok[i]) == o_ok and dct[i].tobytes() == o_ct and np.array_equal(dz[i], o_z), i
        assert np.array_equal(dh[i], o_h if o_ok else np.zeros_like(o_h)), i
        n_ok += o_ok
    assert 64 <= n_ok < 256
    # the encoder's contract (encodings.rs:249-250)
    zbad = z[:2].copy()
    zbad[0, 0, 5] = -p.gamma1
    zbad[1, p.l - 1, 255] = p.gamma1 + 1
    _, ok2 = hp.sig_encode(pset, dev(ct[:2]), dev(zbad), dev(h[:2]))
    assert not host(ok2).any()


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_signer_output_through_sig_decode(hp, pset):
    """What mldsa_sign wrote, taken apart by mldsa_sig_decode: equal to the oracle's sig_decode of the oracle's signature, ||z||inf
    below gamma1 - beta, at most omega hints, and mldsa_sig_encode puts the same bytes back."""
    from fips204_amd.ml_dsa import MlDsa
    p = orc.params(pset)
    m = MlDsa(pset, hotpath=hp)
    rng = np.random.default_rng(5 + pset)
    n = 32
    xi = rng.integers(0, 256, (4, 32), dtype=np.uint8)
    pk, sk = m.keygen_host(xi)
    msgs = [rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8).tobytes() for _ in range(n)]
    rnd = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    kidx = (np.arange(n) % 4).astype(np.uint32)
    sig = np.ascontiguousarray(m.sign_host(sk, msgs, rnd, key_idx=kidx))
    ct, z, h, ok = (host(t) for t in hp.sig_decode(pset, dev(sig)))
    assert ok.all() and np.abs(z).max() < p.gamma1 - p.beta and h.reshape(n, -1).sum(1).max() <= p.omega
    for i in range(n):
        o_ok, o_ct, o_z, o_h = orc.sig_decode(pset, sig[i].tobytes())
        assert o_ok and ct[i].tobytes() == o_ct and np.array_equal(z[i], o_z) and np.array_equal(h[i], o_h)
    again, ok2 = hp.sig_encode(pset, dev(ct), dev(z), dev(h))
    assert host(ok2).all() and np.array_equal(host(again), sig)


@pytest.mark.parametrize("pset", [44, 65, 87])
def test_w1_encode(hp, pset):
    """w1_encode (encodings.rs:338-360): 6-bit fields for gamma2 = (q-1)/88 (values 0..43), 4-bit for (q-1)/32 (0..15)"""
    p = orc.params(pset)
    m = (orc.Q - 1) // (2 * p.gamma2)
    rng = np.random.default_rng(9 + pset)
    n = 40
    w1 = rng.integers(0, m, (n, p.k, N)).astype(np.int32)
    w1[0] = 0
    w1[1] = m - 1
    w1[2] = (np.arange(p.k * N) % m).reshape(p.k, N)
    got = host(hp.w1_encode(pset, dev(w1)))
    assert got.shape == (n, p.w1_len)
    for i in range(n):
        assert got[i].tobytes() == orc.w1_encode(pset, w1[i]), i
    # the same bytes through the generic packer: w1_encode IS simple_bit_pack per polynomial (encodings.rs:354-358)
    assert np.array_equal(host(hp.bit_pack(dev(w1.reshape(-1, N)), 0, m - 1)).reshape(n, -1), got)


@pytest.mark.parametrize("bits", [128, 256])
def test_xof_seam(hp, bits):
    """h256_xof / g128_xof (hashing.rs:13-27; SURVEY row A12: the `sha3` crate's SHAKE, pinned here by hashlib): every input length
    0 .. 2 x rate + 9 (all padding positions, the 0x1F | 0x80 byte at rate - 1), a few long ones, output lengths inside one block,
    at the block boundary and over several blocks; a malformed offset pair refuses only its own op."""
    import hashlib
    rate = 168 if bits == 128 else 136
    fn = hashlib.shake_128 if bits == 128 else hashlib.shake_256
    rng = np.random.default_rng(bits)
    lens = list(range(0, 2 * rate + 10)) + [1000, 1312, 2592, 4627, 7727]
    items = [rng.integers(0, 256, n, dtype=np.uint8).tobytes() for n in lens]
    flat = np.frombuffer(b"".join(items), dtype=np.uint8)
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    d_flat, d_off = dev(flat), torch.from_numpy(off.view(np.int64)).cuda()
    for out_len in (1, 32, 64, rate - 1, rate, rate + 1, 640, 3 * rate + 5):
        out, bad = hp.xof(bits, d_flat, d_off, out_len)
        out = host(out)
        assert not host(bad).any()
        for i, m in enumerate(items):
            assert out[i].tobytes() == fn(m).digest(out_len), (len(m), out_len)
    # the oracle's own sponge agrees (it is what every other parity test leans on)
    assert orc.shake(bits, items[200], 99) == fn(items[200]).digest(99)
    # untrusted offsets: op 3 decreasing, op 7 past the end; their neighbours unaffected
    evil = off.copy()
    evil[4] = evil[3] - 1 if evil[3] else 0
    evil[4] = 2
    evil[8] = np.uint64(2 ** 63)
    out, bad = hp.xof(bits, d_flat, torch.from_numpy(evil.view(np.int64)).cuda(), 32)
    out, bad = host(out), host(bad)
    for i in range(len(items)):
        lo, hi = int(evil[i]), int(evil[i + 1])
        ok_pair = int(evil[0]) <= lo <= hi <= int(evil[-1])
        assert bool(bad[i]) == (not ok_pair), i
        assert out[i].tobytes() == (fn(flat[lo:hi].tobytes()).digest(32) if ok_pair else bytes(32)), i
    assert bad.sum() >= 2
    p = C.c_void_p(d_off.data_ptr())
    assert hp.lib.mldsa_xof(hp._h, 512, p, p, p, 32, None, 1, None) == _lib.ERR_PARAM
    assert hp.lib.mldsa_xof(hp._h, 256, None, None, None, 32, None, 0, None) == _lib.OK
