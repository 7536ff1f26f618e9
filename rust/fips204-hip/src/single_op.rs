//! The reference's own traits, ONE operation per call, on the device: `impl Verifier / Signer / KeyGen` that forward to a
//! process-wide batcher (`mldsa_batcher_*`, csrc/batcher.cpp).
//!
//! What `benches/benchmark.rs:28-62` times and what existing users of the crate call stays exactly that call shape
//! (src/traits.rs:118-308, 330-362).  Any number of threads may call at once: the batcher coalesces the calls in flight into
//! one batched library call, keeps what `try_from_bytes` makes of a key -- and A_hat = ExpandA(rho), the pre-compute
//! `benches/README.md:4-8` names -- in a device-resident table found again by the key's bytes, and wakes every caller with its own
//! result.  A lone caller is better served by the CPU body (28 us a verification, benches/README.md:25, against ~100 us for a
//! one-operation device call): `Shim::cpu_below` keeps that path for a process that is not under load.
//!
//! The newtypes hold the WIRE bytes (what the batcher's key table is indexed by) next to the reference's own key object, so the
//! CPU body and every other method of the reference type stay available through `Deref`.

use std::sync::{Arc, OnceLock};

use fips204::traits::{KeyGen, SerDes, Signer, Verifier};
use fips204::Ph;
use rand_core::CryptoRngCore;
use sha2::{Digest, Sha256, Sha512};
use sha3::digest::{ExtendableOutput, Update, XofReader};
use sha3::Shake128;

use crate::{check, sys, ParamSet};

/// `OID || PH(M)`: what HashML-DSA signs in place of the message (FIPS 204 Algorithm 4 lines 10-22; the reference's crate-private
/// `hashing::hash_message`, src/hashing.rs:317-354, restated on the same `sha2` / `sha3` crates).  PH(M) is message-length-bound host
/// work; the device receives these 43 or 75 bytes as the message of a `MLDSA_MODE_PREHASH` operation and builds
/// `M' = 1 || |ctx| || ctx || OID || PH(M)` itself (csrc `k_mu`), exactly as `fips204_amd/host/fips204_hip.hpp hash_message` does.
pub fn hash_message(message: &[u8], ph: &Ph) -> ([u8; 75], usize) {
    let mut out = [0u8; 75];
    out[..10].copy_from_slice(&[0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02]);
    let len = match ph {
        Ph::SHA256 => {
            out[10] = 0x01;
            out[11..43].copy_from_slice(&Sha256::digest(message));
            43
        }
        Ph::SHA512 => {
            out[10] = 0x03;
            out[11..75].copy_from_slice(&Sha512::digest(message));
            75
        }
        Ph::SHAKE128 => {
            out[10] = 0x0B;
            let mut hasher = Shake128::default();
            hasher.update(message);
            hasher.finalize_xof().read(&mut out[11..43]);
            43
        }
    };
    (out, len)
}

/// One batcher per parameter set and process: `mldsa_batcher_create_on(devices, set, max_batch, max_wait_us, cache_keys)`.
pub struct Batcher {
    raw: *mut sys::mldsa_batcher,
}
unsafe impl Send for Batcher {}
unsafe impl Sync for Batcher {} // mldsa_batcher_verify / _sign / _keygen may be called from any number of threads

impl Batcher {
    /// `devices`: one dispatcher ("lane") per entry; a GPU may be listed twice (two lanes overlap small batches on one device).
    pub fn new(devices: &[i32], set: ParamSet, max_batch: usize, max_wait_us: u32, cache_keys: usize) -> Result<Self, &'static str> {
        let mut raw = core::ptr::null_mut();
        check(unsafe { sys::mldsa_batcher_create_on(devices.as_ptr(), devices.len() as i32, set.id(), max_batch, max_wait_us, cache_keys, &mut raw) })?;
        Ok(Batcher { raw })
    }
    pub fn stats(&self) -> sys::mldsa_batcher_stats {
        let mut s = sys::mldsa_batcher_stats { batches: 0, requests: 0, largest_batch: 0, keys_expanded: 0, key_hits: 0 };
        unsafe { sys::mldsa_batcher_get_stats(self.raw, &mut s) };
        s
    }
    /// Drop one private key from the device-resident table (the caller is about to zeroize its own copy), or every key.
    pub fn forget_key(&self, sk: &[u8]) -> Result<(), &'static str> {
        check(unsafe { sys::mldsa_batcher_forget_key(self.raw, sk.as_ptr(), sk.len()) })
    }
    pub fn flush_keys(&self) -> Result<(), &'static str> {
        check(unsafe { sys::mldsa_batcher_flush_keys(self.raw) })
    }
}

impl Drop for Batcher {
    fn drop(&mut self) {
        unsafe { sys::mldsa_batcher_destroy(self.raw) }
    }
}

macro_rules! single_op_shim {
    ($modname:ident, $refmod:path, $set:expr) => {
        pub mod $modname {
            use super::*;
            use $refmod as reference;

            pub const PK_LEN: usize = reference::PK_LEN;
            pub const SK_LEN: usize = reference::SK_LEN;
            pub const SIG_LEN: usize = reference::SIG_LEN;

            static BATCHER: OnceLock<Batcher> = OnceLock::new();

            /// The process-wide batcher of this parameter set: device 0, batches of up to 4 096 operations, no artificial wait, a
            /// key table of the library's default size.  Call `init` first to choose devices and limits.
            pub fn batcher() -> &'static Batcher {
                BATCHER.get_or_init(|| Batcher::new(&[0], $set, 4096, 0, 0).expect("mldsa_batcher_create_on"))
            }
            /// The batcher if one exists already (never creates one: `Drop` must not initialise a device).
            pub fn existing_batcher() -> Option<&'static Batcher> {
                BATCHER.get()
            }
            pub fn init(devices: &[i32], max_batch: usize, max_wait_us: u32, cache_keys: usize) -> Result<(), &'static str> {
                let b = Batcher::new(devices, $set, max_batch, max_wait_us, cache_keys)?;
                BATCHER.set(b).map_err(|_| "fips204-hip: the batcher of this parameter set already exists")
            }

            /// `fips204::ml_dsa_NN::PublicKey` + its wire bytes.
            #[derive(Clone)]
            pub struct HipPublicKey {
                inner: reference::PublicKey,
                wire: [u8; PK_LEN],
            }
            impl core::ops::Deref for HipPublicKey {
                type Target = reference::PublicKey;
                fn deref(&self) -> &Self::Target {
                    &self.inner
                }
            }
            impl SerDes for HipPublicKey {
                type ByteArray = [u8; PK_LEN];
                fn try_from_bytes(bytes: Self::ByteArray) -> Result<Self, &'static str> {
                    Ok(HipPublicKey { inner: reference::PublicKey::try_from_bytes(bytes)?, wire: bytes })
                }
                fn into_bytes(self) -> Self::ByteArray {
                    self.wire
                }
            }
            impl Verifier for HipPublicKey {
                type Signature = [u8; SIG_LEN];
                /// src/lib.rs:364-380, signature unchanged; every failure is `false` (lib.rs:368-370)
                fn verify(&self, message: &[u8], sig: &Self::Signature, ctx: &[u8]) -> bool {
                    let mut ok = 0u8;
                    let rc = unsafe {
                        sys::mldsa_batcher_verify(batcher().raw, sys::MLDSA_MODE_PURE, self.wire.as_ptr(), message.as_ptr(), message.len(),
                                                  ctx.as_ptr(), ctx.len(), sig.as_ptr(), &mut ok)
                    };
                    rc == sys::MLDSA_OK && ok != 0
                }
                /// HashML-DSA.Verify (src/lib.rs:391-411): PH(M) on the host (`hash_message`), then ONE device operation whose message
                /// is `OID || PH(M)` in `MLDSA_MODE_PREHASH`; a ctx longer than 255 bytes is `false` (lib.rs:395-397; the device
                /// refuses it too).
                fn hash_verify(&self, message: &[u8], sig: &Self::Signature, ctx: &[u8], ph: &Ph) -> bool {
                    if ctx.len() > 255 {
                        return false;
                    }
                    let (m, m_len) = hash_message(message, ph);
                    let mut ok = 0u8;
                    let rc = unsafe {
                        sys::mldsa_batcher_verify(batcher().raw, sys::MLDSA_MODE_PREHASH, self.wire.as_ptr(), m.as_ptr(), m_len, ctx.as_ptr(),
                                                  ctx.len(), sig.as_ptr(), &mut ok)
                    };
                    rc == sys::MLDSA_OK && ok != 0
                }
            }

            /// The wire bytes of a private key, shared by every clone of the key object: the LAST owner to go takes the key out of the
            /// device-resident table and clears the bytes (the reference's key is `ZeroizeOnDrop`, src/types.rs:19).  Never creates a
            /// batcher (a key parsed and dropped on a host without a usable GPU must not panic inside `drop`): if none exists, no
            /// device copy exists either.
            struct SkWire([u8; SK_LEN]);
            impl Drop for SkWire {
                fn drop(&mut self) {
                    if let Some(b) = existing_batcher() {
                        let _ = b.forget_key(&self.0);
                    }
                    for b in self.0.iter_mut() {
                        unsafe { core::ptr::write_volatile(b, 0) };
                    }
                }
            }

            /// `fips204::ml_dsa_NN::PrivateKey` + its wire bytes.
            #[derive(Clone)]
            pub struct HipPrivateKey {
                inner: reference::PrivateKey,
                wire: Arc<SkWire>,
            }
            impl SerDes for HipPrivateKey {
                type ByteArray = [u8; SK_LEN];
                fn try_from_bytes(bytes: Self::ByteArray) -> Result<Self, &'static str> {
                    Ok(HipPrivateKey { inner: reference::PrivateKey::try_from_bytes(bytes)?, wire: Arc::new(SkWire(bytes)) })
                }
                fn into_bytes(self) -> Self::ByteArray {
                    self.wire.0 // (a copy for the caller; the shared bytes are cleared when their last owner goes)
                }
            }
            impl HipPrivateKey {
                /// steps 1-9 and 23-25 of HashML-DSA.Sign (src/lib.rs:310-342) around ONE device operation in `MLDSA_MODE_PREHASH`
                fn hash_sign_on_device(&self, rnd: &[u8; 32], message: &[u8], ctx: &[u8], ph: &Ph) -> Result<[u8; SIG_LEN], &'static str> {
                    if ctx.len() > 255 {
                        return Err("HashML-DSA.Sign: ctx too long");
                    }
                    let (m, m_len) = hash_message(message, ph);
                    let mut sig = [0u8; SIG_LEN];
                    check(unsafe {
                        sys::mldsa_batcher_sign(batcher().raw, sys::MLDSA_MODE_PREHASH, self.wire.0.as_ptr(), m.as_ptr(), m_len, ctx.as_ptr(),
                                                ctx.len(), rnd.as_ptr(), sig.as_mut_ptr())
                    })?;
                    Ok(sig)
                }
            }
            impl Signer for HipPrivateKey {
                type Signature = [u8; SIG_LEN];
                type PublicKey = HipPublicKey;

                /// src/lib.rs:268-296: rnd from the caller's RNG exactly as the reference draws it (lib.rs:282-283)
                fn try_sign_with_rng(&self, rng: &mut impl CryptoRngCore, message: &[u8], ctx: &[u8]) -> Result<Self::Signature, &'static str> {
                    let mut rnd = [0u8; 32];
                    rng.try_fill_bytes(&mut rnd).map_err(|_| "Random number generator failed")?;
                    self.try_sign_with_seed(&rnd, message, ctx)
                }
                fn try_sign_with_seed(&self, rnd: &[u8; 32], message: &[u8], ctx: &[u8]) -> Result<Self::Signature, &'static str> {
                    let mut sig = [0u8; SIG_LEN];
                    check(unsafe {
                        sys::mldsa_batcher_sign(batcher().raw, sys::MLDSA_MODE_PURE, self.wire.0.as_ptr(), message.as_ptr(), message.len(),
                                                ctx.as_ptr(), ctx.len(), rnd.as_ptr(), sig.as_mut_ptr())
                    })?;
                    Ok(sig)
                }
                /// src/lib.rs:310-342 on the device: the ctx check comes first, then rnd from the caller's RNG, as in the reference
                fn try_hash_sign_with_rng(&self, rng: &mut impl CryptoRngCore, message: &[u8], ctx: &[u8], ph: &Ph) -> Result<Self::Signature, &'static str> {
                    if ctx.len() > 255 {
                        return Err("HashML-DSA.Sign: ctx too long");
                    }
                    let mut rnd = [0u8; 32];
                    rng.try_fill_bytes(&mut rnd).map_err(|_| "HashML-DSA.Sign: random number generator failed")?;
                    self.hash_sign_on_device(&rnd, message, ctx, ph)
                }
                fn try_hash_sign_with_seed(&self, rnd: &[u8; 32], message: &[u8], ctx: &[u8], ph: &Ph) -> Result<Self::Signature, &'static str> {
                    self.hash_sign_on_device(rnd, message, ctx, ph)
                }
                /// src/lib.rs:345-349
                fn get_public_key(&self) -> Self::PublicKey {
                    let pk = self.inner.get_public_key();
                    let wire = pk.clone().into_bytes();
                    HipPublicKey { inner: pk, wire }
                }
            }

            /// `fips204::ml_dsa_NN::KG` on the device.
            pub struct HipKeyGen;
            impl KeyGen for HipKeyGen {
                type PublicKey = HipPublicKey;
                type PrivateKey = HipPrivateKey;

                fn try_keygen_with_rng(rng: &mut impl CryptoRngCore) -> Result<(Self::PublicKey, Self::PrivateKey), &'static str> {
                    let mut xi = [0u8; 32];
                    rng.try_fill_bytes(&mut xi).map_err(|_| "Random number generator failed")?; // src/lib.rs:236-238
                    Ok(Self::keygen_from_seed(&xi))
                }
                /// src/lib.rs:247-250
                fn keygen_from_seed(xi: &[u8; 32]) -> (Self::PublicKey, Self::PrivateKey) {
                    let (mut pk, mut sk) = ([0u8; PK_LEN], [0u8; SK_LEN]);
                    let rc = unsafe { sys::mldsa_batcher_keygen(batcher().raw, xi.as_ptr(), pk.as_mut_ptr(), sk.as_mut_ptr()) };
                    assert_eq!(rc, sys::MLDSA_OK, "mldsa_batcher_keygen");
                    (HipPublicKey::try_from_bytes(pk).expect("device keygen produced a malformed public key"),
                     HipPrivateKey::try_from_bytes(sk).expect("device keygen produced a malformed private key"))
                }
            }
        }
    };
}

single_op_shim!(ml_dsa_44, fips204::ml_dsa_44, ParamSet::MlDsa44);
single_op_shim!(ml_dsa_65, fips204::ml_dsa_65, ParamSet::MlDsa65);
single_op_shim!(ml_dsa_87, fips204::ml_dsa_87, ParamSet::MlDsa87);
