#!/usr/bin/env python3
"""Write rust/fips204-hip-sys/src/lib.rs from include/mldsa_hip.h (what bindgen would do; neither bindgen nor rustc is in this image).

    python tools/gen_rust_sys.py          regenerate the file
    python tools/gen_rust_sys.py --check  exit 1 when the committed file differs from what the header gives

Every prototype, every #[repr(C)] struct and every MLDSA_* constant of the header, one to one, in the header's order.
tests/test_rust_binding_cpu.py checks the result against the header with parsers of its own."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cheader  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mldsa_hip.h")
OUT = os.path.join(ROOT, "rust", "fips204-hip-sys", "src", "lib.rs")

SCALARS = {"int": "c_int", "unsigned": "c_uint", "unsigned int": "c_uint", "long": "c_long", "unsigned long long": "c_ulonglong", "size_t": "usize",
           "uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "char": "c_char", "void": "c_void"}
RUST_KEYWORDS = {"in", "type", "ref", "box", "fn", "mod", "use", "as", "loop", "match", "move", "priv", "pub", "self", "super", "where", "yield"}


def rust_type(ctype):
    """'const uint8_t *' -> '*const u8'; 'mldsa_ctx **' -> '*mut *mut mldsa_ctx'; 'uint8_t * const *' -> '*const *mut u8'"""
    toks = ctype.replace("*", " * ").split()
    # base type = everything up to the first '*', minus const
    base, i, base_const = [], 0, False
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            base_const = True
        else:
            base.append(toks[i])
        i += 1
    base = " ".join(base)
    t = SCALARS.get(base, base)
    pointee_const = base_const
    while i < len(toks):
        assert toks[i] == "*", ctype
        t = ("*const " if pointee_const else "*mut ") + t
        pointee_const = False
        i += 1
        if i < len(toks) and toks[i] == "const":
            pointee_const = True
            i += 1
    return t


def ident(name):
    return name + "_" if name in RUST_KEYWORDS else name


def generate():
    h = cheader.parse(HEADER)
    out = []
    w = out.append
    w("//! Raw FFI binding of `libmldsa_hip.so` (include/mldsa_hip.h): the MI355X batched ML-DSA hot path behind a C ABI.")
    w("//!")
    w("//! GENERATED from the header by tools/gen_rust_sys.py -- do not edit; tests/test_rust_binding_cpu.py pins it to the header")
    w("//! (names, arity, argument classes, struct field order and widths, constants).  This crate is the ONE place with `unsafe`")
    w("//! declarations, so that `#![deny(unsafe_code)]` (fips204 src/lib.rs:2) keeps holding for the reference crate itself; the seams it")
    w("//! stands behind are the crate-private imports of src/ml_dsa.rs:3-11 and the constants of src/lib.rs:118-124.")
    w("#![no_std]")
    w("#![allow(non_camel_case_types)]")
    w("use core::ffi::{c_char, c_int, c_long, c_uint, c_ulonglong, c_void};")
    w("")
    for name in h["opaque"]:
        w("#[repr(C)]")
        w(f"pub struct {name} {{")
        w("    _private: [u8; 0],")
        w("}")
    w("")
    for name, value in h["consts"]:
        w(f"pub const {name}: c_int = {value};")
    w("")
    for name, fields in h["structs"]:
        w("#[repr(C)]")
        w("#[derive(Clone, Copy, Debug)]")
        w(f"pub struct {name} {{")
        for ctype, fname in fields:
            w(f"    pub {ident(fname)}: {rust_type(ctype)},")
        w("}")
    w("")
    w('#[link(name = "mldsa_hip")]')
    w('extern "C" {')
    for name, ret, args in h["funcs"]:
        a = ", ".join(f"{ident(n)}: {rust_type(t)}" for t, n in args)
        r = "" if ret == "void" else f" -> {rust_type(ret)}"
        line = f"    pub fn {name}({a}){r};"
        if len(line) > 150:  # wrap long prototypes at argument boundaries
            parts, cur = [], f"    pub fn {name}("
            for k, (t, n) in enumerate(args):
                piece = f"{ident(n)}: {rust_type(t)}" + (", " if k + 1 < len(args) else "")
                if len(cur) + len(piece) > 150:
                    parts.append(cur.rstrip())
                    cur = "        " + piece
                else:
                    cur += piece
            parts.append(cur + f"){r};")
            line = "\n".join(parts)
        w(line)
    w("}")
    return "\n".join(out) + "\n"


def main():
    text = generate()
    if "--check" in sys.argv:
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("rust/fips204-hip-sys/src/lib.rs is stale: run python tools/gen_rust_sys.py", file=sys.stderr)
            return 1
        return 0
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {text.count('pub fn ')} functions")
    return 0


if __name__ == "__main__":
    sys.exit(main())
