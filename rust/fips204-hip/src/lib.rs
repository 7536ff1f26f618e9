//! `fips204-hip`: integritychain/fips204's trait surface on the MI355X library.
//!
//! The reference crate is `#![deny(unsafe_code)]` (src/lib.rs:2) and single-operation; the device wants batches.  This crate
//! sits beside it and offers two front-ends over `fips204-hip-sys`:
//!
//! * [`batch`] -- `verify_many` / `sign_many` / `keygen_many` on slices of wire-format keys (`SerDes::into_bytes`,
//!   src/traits.rs:330-362 for the single-op shape these replace): one library call per batch, host memory in and out.
//! * [`single_op`] -- `HipPublicKey` / `HipPrivateKey` / `HipKeyGen` implementing the reference's own `Verifier` / `Signer` /
//!   `KeyGen` traits (src/traits.rs:118-308) ONE operation per call: every call goes to a process-wide batcher
//!   (`mldsa_batcher_*`), which coalesces the calls of all threads in flight and keeps each key's expanded form and A_hat in a
//!   device-resident table.
//!
//! Nothing here is built in the repository's CI image (no rustc there); `tests/test_rust_binding_cpu.py` pins the `-sys` crate
//! to the header, and the same ABI is exercised by the C, C++ and ctypes hosts of the test-suite.
#![allow(clippy::too_many_arguments)]

pub mod batch;
pub mod single_op;

pub use fips204_hip_sys as sys;

use core::ffi::CStr;
use core::ptr;

/// Parameter sets (src/lib.rs:118-124 instantiates the three modules from these constants).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub enum ParamSet {
    MlDsa44,
    MlDsa65,
    MlDsa87,
}

impl ParamSet {
    pub const fn id(self) -> i32 {
        match self {
            ParamSet::MlDsa44 => sys::MLDSA_44,
            ParamSet::MlDsa65 => sys::MLDSA_65,
            ParamSet::MlDsa87 => sys::MLDSA_87,
        }
    }
    /// PK_LEN, SK_LEN, SIG_LEN as the library reports them (mldsa_get_params): equal to the reference's constants
    pub fn lengths(self) -> (usize, usize, usize) {
        let mut p = core::mem::MaybeUninit::<sys::mldsa_params>::zeroed();
        let rc = unsafe { sys::mldsa_get_params(self.id(), p.as_mut_ptr()) };
        assert_eq!(rc, sys::MLDSA_OK);
        let p = unsafe { p.assume_init() };
        (p.pk_len as usize, p.sk_len as usize, p.sig_len as usize)
    }
}

/// The library's status codes as the crate's `Result<_, &'static str>` (src/helpers.rs:12-18 `ensure!`).
pub fn check(rc: i32) -> Result<(), &'static str> {
    match rc {
        sys::MLDSA_OK => Ok(()),
        sys::MLDSA_ERR_PARAM => Err("mldsa_hip: bad parameter (unknown set, NULL pointer, malformed offsets, key index out of range)"),
        sys::MLDSA_ERR_CTX_LEN => Err("ML-DSA.Sign: ctx too long"), // src/lib.rs:274
        sys::MLDSA_ERR_DEVICE => Err("mldsa_hip: HIP runtime error (see last_error())"),
        sys::MLDSA_ERR_NOMEM => Err("mldsa_hip: device workspace allocation failed"),
        sys::MLDSA_ERR_AGAIN => Err("mldsa_hip: operation left unfinished by an asynchronous signing call"),
        _ => Err("mldsa_hip: unknown status"),
    }
}

/// Text of the calling thread's last failure.
pub fn last_error() -> String {
    let p = unsafe { sys::mldsa_last_error() };
    if p.is_null() {
        return String::new();
    }
    unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
}

/// One device context (twiddle tables, workspace, streams).  `Send` but not `Sync`: op-level calls of one context are serialised
/// by the caller, exactly like the C ABI says; use one `Context` per thread, a [`Group`] or the batcher.
pub struct Context {
    raw: *mut sys::mldsa_ctx,
}
unsafe impl Send for Context {}

impl Context {
    pub fn new(device_id: i32) -> Result<Self, &'static str> {
        if unsafe { sys::mldsa_abi_version() } != sys::MLDSA_ABI_VERSION {
            return Err("mldsa_hip: the library's ABI version differs from the one this binding was generated from");
        }
        let mut raw = ptr::null_mut();
        check(unsafe { sys::mldsa_ctx_create(device_id, &mut raw) })?;
        Ok(Context { raw })
    }
    pub fn raw(&self) -> *mut sys::mldsa_ctx {
        self.raw
    }
    /// Size the workspace ahead of time so that no later call of up to `n_ops` operations waits for an allocation.
    pub fn reserve(&self, set: ParamSet, op: i32, n_ops: usize) -> Result<(), &'static str> {
        check(unsafe { sys::mldsa_reserve(self.raw, set.id(), op, n_ops) })
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::mldsa_ctx_destroy(self.raw) }
    }
}

/// All GPUs of a node from one host thread: one context + worker thread per device, contiguous ceil(B / N) slices, no collective
/// on the data path (BASELINE config 4 = `Group::new(&[0, 1, 2, 3, 4, 5, 6, 7])`).
pub struct Group {
    raw: *mut sys::mldsa_group,
}
unsafe impl Send for Group {}

impl Group {
    pub fn new(devices: &[i32]) -> Result<Self, &'static str> {
        let mut raw = ptr::null_mut();
        check(unsafe { sys::mldsa_group_create(devices.as_ptr(), devices.len() as i32, &mut raw) })?;
        Ok(Group { raw })
    }
    pub fn raw(&self) -> *mut sys::mldsa_group {
        self.raw
    }
    pub fn len(&self) -> usize {
        unsafe { sys::mldsa_group_size(self.raw) as usize }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
    /// Slices that already live on the devices (keys expanded there, inputs uploaded there): `slices[i]` = the arguments of
    /// `mldsa_verify` for device i.  `wait = false` returns once everything is enqueued; [`Group::sync`] waits later.
    pub fn verify_resident(&self, set: ParamSet, mode: i32, slices: &[sys::mldsa_verify_slice], wait: bool) -> Result<(), &'static str> {
        assert_eq!(slices.len(), self.len());
        check(unsafe { sys::mldsa_verify_group(self.raw, set.id(), mode, slices.as_ptr(), wait as i32) })
    }
    pub fn sign_resident(&self, set: ParamSet, mode: i32, slices: &[sys::mldsa_sign_slice], wait: bool) -> Result<(), &'static str> {
        assert_eq!(slices.len(), self.len());
        check(unsafe { sys::mldsa_sign_group(self.raw, set.id(), mode, slices.as_ptr(), wait as i32) })
    }
    pub fn keygen_resident(&self, set: ParamSet, slices: &[sys::mldsa_keygen_slice], wait: bool) -> Result<(), &'static str> {
        assert_eq!(slices.len(), self.len());
        check(unsafe { sys::mldsa_keygen_group(self.raw, set.id(), slices.as_ptr(), wait as i32) })
    }
    pub fn sync(&self) -> Result<(), &'static str> {
        check(unsafe { sys::mldsa_group_sync(self.raw) })
    }
}

impl Drop for Group {
    fn drop(&mut self) {
        unsafe { sys::mldsa_group_destroy(self.raw) }
    }
}

/// `msgs` back to back plus the n + 1 offsets the C ABI takes (`off[i] .. off[i + 1]` = item i).
pub fn concat_with_offsets(items: &[&[u8]]) -> (Vec<u8>, Vec<u64>) {
    let mut buf = Vec::with_capacity(items.iter().map(|m| m.len()).sum());
    let mut off = Vec::with_capacity(items.len() + 1);
    off.push(0u64);
    for m in items {
        buf.extend_from_slice(m);
        off.push(buf.len() as u64);
    }
    (buf, off)
}
