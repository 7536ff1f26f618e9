"""Argument validation of the host-memory mirror (fips204_amd/ml_dsa.py verify_host / sign_host / keygen_host): a mismatch
between the number of operations and the messages / ctxs / offsets / output buffers must be a ValueError raised before the
C ABI is entered -- the library walks n_ops + 1 offsets and writes n_ops results, so a short array would be a heap
out-of-bounds access on the C side (the reference's slices carry their own lengths, src/traits.rs:330-362)."""
import numpy as np
import pytest

from fips204_amd.ml_dsa import MlDsa


def test_string_lists_must_match_the_batch():
    flat, off = MlDsa._host_strings([b"ab", b"", b"cde"], 3, "messages")
    assert off.dtype == np.uint64 and off.tolist() == [0, 2, 2, 5] and flat[:5].tobytes() == b"abcde"
    with pytest.raises(ValueError, match="2 entries for 3"):
        MlDsa._host_strings([b"ab", b"c"], 3, "messages")
    with pytest.raises(ValueError, match="4 entries for 3"):
        MlDsa._host_strings([b""] * 4, 3, "ctxs")


def test_flat_offset_pairs_are_checked():
    flat = np.frombuffer(b"abcdef", dtype=np.uint8)
    ok_off = np.array([0, 2, 6], dtype=np.uint64)
    f, o = MlDsa._host_strings((flat, ok_off), 2, "messages")
    assert o is ok_off or np.array_equal(o, ok_off)
    # int64 offsets are converted, not reinterpreted
    f, o = MlDsa._host_strings((flat, np.array([0, 2, 6], dtype=np.int64)), 2, "messages")
    assert o.dtype == np.uint64 and o.tolist() == [0, 2, 6]
    with pytest.raises(ValueError, match="n_ops \\+ 1"):
        MlDsa._host_strings((flat, np.array([0, 2], dtype=np.uint64)), 2, "messages")        # 9 messages for 10 signatures
    with pytest.raises(ValueError, match="past the end"):
        MlDsa._host_strings((flat, np.array([0, 2, 7], dtype=np.uint64)), 2, "messages")     # truncated byte buffer
    with pytest.raises(ValueError, match="non-decreasing"):
        MlDsa._host_strings((flat, np.array([0, 4, 2], dtype=np.uint64)), 2, "messages")
    with pytest.raises(ValueError, match="non-negative"):
        MlDsa._host_strings((flat, np.array([0, -1, 2], dtype=np.int64)), 2, "messages")
    # a non-contiguous flat buffer is made contiguous (the C side reads it linearly)
    f, o = MlDsa._host_strings((np.arange(12, dtype=np.uint8)[::2], np.array([0, 3, 6], dtype=np.uint64)), 2, "messages")
    assert f.flags.c_contiguous and f.tolist() == [0, 2, 4, 6, 8, 10]


def test_output_buffers_are_checked():
    assert MlDsa._host_out(np.zeros(8, np.uint8), np.uint8, 8, "out") is not None
    for bad in (np.zeros(7, np.uint8),                 # short
                np.zeros(8, np.int8),                  # wrong dtype
                np.zeros(16, np.uint8)[::2],           # not contiguous
                [0] * 8):                              # not an array
        with pytest.raises(ValueError):
            MlDsa._host_out(bad, np.uint8, 8, "out")
    ro = np.zeros(8, np.uint8)
    ro.flags.writeable = False
    with pytest.raises(ValueError):
        MlDsa._host_out(ro, np.uint8, 8, "out")
