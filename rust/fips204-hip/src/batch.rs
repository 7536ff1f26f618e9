//! Batch front-end: many independent operations per call, wire-format keys, host memory in and out.
//!
//! The single-operation shapes these replace are `Verifier::verify`, `Signer::try_sign_with_seed` and `KeyGen::keygen_from_seed`
//! (src/traits.rs:330-362, 118-308; bodies in src/lib.rs:247-296, 364-380).  Keys travel as the byte arrays `SerDes::into_bytes`
//! produces; `key_idx[op]` names the key of operation `op`, so a table of a few keys serves a large batch and is deserialised
//! (`try_from_bytes`, src/ml_dsa.rs:445-498) once per call.  Results are bit for bit those of the reference: a failed verification is
//! `false` (src/lib.rs:368-370), a refused signing operation an `Err` in its own slot.

use crate::{check, concat_with_offsets, sys, Context, Group, ParamSet};

/// `Verifier::verify` for a batch.  `pks`: wire-format public keys; `key_idx[i] < pks.len()`; `ctxs[i]` at most 255 bytes.
pub fn verify_many<const PK_LEN: usize, const SIG_LEN: usize>(
    cx: &Context, set: ParamSet, pks: &[[u8; PK_LEN]], key_idx: &[u32], msgs: &[&[u8]], sigs: &[[u8; SIG_LEN]], ctxs: &[&[u8]],
) -> Result<Vec<bool>, &'static str> {
    let n = sigs.len();
    assert!(key_idx.len() == n && msgs.len() == n && ctxs.len() == n);
    let (msg_buf, msg_off) = concat_with_offsets(msgs);
    let (ctx_buf, ctx_off) = concat_with_offsets(ctxs);
    let mut ok = vec![0u8; n];
    check(unsafe {
        sys::mldsa_verify_host(cx.raw(), set.id(), sys::MLDSA_MODE_PURE, pks.as_ptr().cast(), pks.len(), key_idx.as_ptr(), msg_buf.as_ptr(),
                               msg_off.as_ptr(), ctx_buf.as_ptr(), ctx_off.as_ptr(), sigs.as_ptr().cast(), ok.as_mut_ptr(), n)
    })?;
    Ok(ok.into_iter().map(|b| b != 0).collect())
}

/// `Signer::try_sign_with_seed` for a batch (`rnd[i]` = the 32 bytes `try_sign_with_rng` draws, src/lib.rs:282-283; all zero for
/// the deterministic variant).  Per-operation results: `Err` for an operation the library refused (ctx too long, bad key index).
pub fn sign_many<const SK_LEN: usize, const SIG_LEN: usize>(
    cx: &Context, set: ParamSet, sks: &[[u8; SK_LEN]], key_idx: &[u32], msgs: &[&[u8]], ctxs: &[&[u8]], rnd: &[[u8; 32]],
) -> Result<Vec<Result<[u8; SIG_LEN], &'static str>>, &'static str> {
    let n = msgs.len();
    assert!(key_idx.len() == n && ctxs.len() == n && rnd.len() == n);
    let (msg_buf, msg_off) = concat_with_offsets(msgs);
    let (ctx_buf, ctx_off) = concat_with_offsets(ctxs);
    let mut sigs = vec![[0u8; SIG_LEN]; n];
    let mut status = vec![0i32; n];
    check(unsafe {
        sys::mldsa_sign_host(cx.raw(), set.id(), sys::MLDSA_MODE_PURE, sks.as_ptr().cast(), sks.len(), key_idx.as_ptr(), msg_buf.as_ptr(),
                             msg_off.as_ptr(), ctx_buf.as_ptr(), ctx_off.as_ptr(), rnd.as_ptr().cast(), sigs.as_mut_ptr().cast(),
                             status.as_mut_ptr(), n)
    })?;
    Ok(sigs.into_iter().zip(status).map(|(s, st)| check(st).map(|_| s)).collect())
}

/// `KeyGen::keygen_from_seed` for a batch: (pk, sk) wire bytes per seed.
pub fn keygen_many<const PK_LEN: usize, const SK_LEN: usize>(
    cx: &Context, set: ParamSet, xi: &[[u8; 32]],
) -> Result<(Vec<[u8; PK_LEN]>, Vec<[u8; SK_LEN]>), &'static str> {
    let n = xi.len();
    let mut pk = vec![[0u8; PK_LEN]; n];
    let mut sk = vec![[0u8; SK_LEN]; n];
    check(unsafe { sys::mldsa_keygen_host(cx.raw(), set.id(), xi.as_ptr().cast(), pk.as_mut_ptr().cast(), sk.as_mut_ptr().cast(), n) })?;
    Ok((pk, sk))
}

/// The same three calls split over the GPUs of a node by the library (contiguous ceil(B / N) slices, byte-identical results).
impl Group {
    pub fn verify_many<const PK_LEN: usize, const SIG_LEN: usize>(
        &self, set: ParamSet, pks: &[[u8; PK_LEN]], key_idx: &[u32], msgs: &[&[u8]], sigs: &[[u8; SIG_LEN]], ctxs: &[&[u8]],
    ) -> Result<Vec<bool>, &'static str> {
        let n = sigs.len();
        let (msg_buf, msg_off) = concat_with_offsets(msgs);
        let (ctx_buf, ctx_off) = concat_with_offsets(ctxs);
        let mut ok = vec![0u8; n];
        check(unsafe {
            sys::mldsa_verify_host_group(self.raw(), set.id(), sys::MLDSA_MODE_PURE, pks.as_ptr().cast(), pks.len(), key_idx.as_ptr(),
                                         msg_buf.as_ptr(), msg_off.as_ptr(), ctx_buf.as_ptr(), ctx_off.as_ptr(), sigs.as_ptr().cast(),
                                         ok.as_mut_ptr(), n)
        })?;
        Ok(ok.into_iter().map(|b| b != 0).collect())
    }

    pub fn sign_many<const SK_LEN: usize, const SIG_LEN: usize>(
        &self, set: ParamSet, sks: &[[u8; SK_LEN]], key_idx: &[u32], msgs: &[&[u8]], ctxs: &[&[u8]], rnd: &[[u8; 32]],
    ) -> Result<Vec<Result<[u8; SIG_LEN], &'static str>>, &'static str> {
        let n = msgs.len();
        let (msg_buf, msg_off) = concat_with_offsets(msgs);
        let (ctx_buf, ctx_off) = concat_with_offsets(ctxs);
        let mut sigs = vec![[0u8; SIG_LEN]; n];
        let mut status = vec![0i32; n];
        check(unsafe {
            sys::mldsa_sign_host_group(self.raw(), set.id(), sys::MLDSA_MODE_PURE, sks.as_ptr().cast(), sks.len(), key_idx.as_ptr(),
                                       msg_buf.as_ptr(), msg_off.as_ptr(), ctx_buf.as_ptr(), ctx_off.as_ptr(), rnd.as_ptr().cast(),
                                       sigs.as_mut_ptr().cast(), status.as_mut_ptr(), n)
        })?;
        Ok(sigs.into_iter().zip(status).map(|(s, st)| check(st).map(|_| s)).collect())
    }

    pub fn keygen_many<const PK_LEN: usize, const SK_LEN: usize>(
        &self, set: ParamSet, xi: &[[u8; 32]],
    ) -> Result<(Vec<[u8; PK_LEN]>, Vec<[u8; SK_LEN]>), &'static str> {
        let n = xi.len();
        let mut pk = vec![[0u8; PK_LEN]; n];
        let mut sk = vec![[0u8; SK_LEN]; n];
        check(unsafe { sys::mldsa_keygen_host_group(self.raw(), set.id(), xi.as_ptr().cast(), pk.as_mut_ptr().cast(), sk.as_mut_ptr().cast(), n) })?;
        Ok((pk, sk))
    }
}
