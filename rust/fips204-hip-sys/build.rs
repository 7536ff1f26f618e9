//! Links libmldsa_hip.so (built by `make -C fips204_amd/csrc`, see the repository's README).
//! MLDSA_HIP_LIB_DIR = the directory that holds the library; the default is the in-tree build directory.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("MLDSA_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../fips204_amd/csrc")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=mldsa_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=MLDSA_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/mldsa_hip.h");
}
