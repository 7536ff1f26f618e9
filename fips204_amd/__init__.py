"""fips204_amd -- MI355X-native batched ML-DSA (FIPS 204) hot path.

Hand-written HIP kernels for gfx950 behind the C ABI of include/mldsa_hip.h; this package
is the thin Python host side (ctypes) that mirrors the reference crate's interface for the
path: `fips204_amd.hotpath` = the crate-private seams (ntt, inv_ntt, mat_vec_mul, expand_a,
...), `fips204_amd.ml_dsa_44 / ml_dsa_65 / ml_dsa_87` = the KeyGen / Signer / Verifier /
SerDes trait surface, batched.
"""
from ._lib import MldsaError, get_params, load  # noqa: F401

__all__ = ["MldsaError", "get_params", "load"]
