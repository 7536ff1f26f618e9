"""The Rust side is pinned to the C header (the binding used to live as fenced text in INTEGRATION.md and had drifted
to 69 of the header's 85 functions).

rust/fips204-hip-sys/src/lib.rs is generated from include/mldsa_hip.h by tools/gen_rust_sys.py; rustc is not in this image, so a textual
comparison is the guard there is.  The parsers below are this test's OWN (regular expressions over both texts, nothing shared with
the generator): same function names, same arity, same pointer / integer / size class and constness per argument and return, same
field order and width for every #[repr(C)] struct, same MLDSA_* constants.  The last tests mutate the header text and check that the
comparison notices.  Seams the binding stands behind: /root/reference/src/ml_dsa.rs:3-11, src/lib.rs:118-124 (and src/lib.rs:2, the
lint that forces a separate -sys crate); trait surface of the shim: src/traits.rs:118-308, 330-362."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mldsa_hip.h")
SYS_RS = os.path.join(ROOT, "rust", "fips204-hip-sys", "src", "lib.rs")

C_SCALAR = {"int": ("int", 4), "unsigned": ("uint", 4), "unsigned int": ("uint", 4), "long": ("long", 8), "unsigned long long": ("u64", 8),
            "size_t": ("usize", 8), "uint8_t": ("u8", 1), "uint16_t": ("u16", 2), "uint32_t": ("u32", 4), "uint64_t": ("u64", 8),
            "int32_t": ("i32", 4), "char": ("char", 1), "void": ("void", 0)}
RS_SCALAR = {"c_int": ("int", 4), "c_uint": ("uint", 4), "c_long": ("long", 8), "c_ulonglong": ("u64", 8), "usize": ("usize", 8), "u8": ("u8", 1),
             "u16": ("u16", 2), "u32": ("u32", 4), "u64": ("u64", 8), "i32": ("i32", 4), "c_char": ("char", 1), "c_void": ("void", 0)}


def c_class(ctype):
    """('ptr', const-ness per level ..., base) -- e.g. 'const uint8_t *' -> ('ptr', 'const', 'u8'); 'int' -> ('int',)"""
    toks = ctype.replace("*", " * ").split()
    base, consts, i, const = [], [], 0, False
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            const = True
        else:
            base.append(toks[i])
        i += 1
    base = " ".join(base)
    kind = C_SCALAR.get(base, (base, None))[0]
    levels = []
    while i < len(toks):
        assert toks[i] == "*"
        levels.append("const" if const else "mut")
        const = False
        i += 1
        if i < len(toks) and toks[i] == "const":
            const = True
            i += 1
    return tuple(["ptr"] * len(levels) + levels[::-1] + [kind]) if levels else (kind,)


def rs_class(rtype):
    toks = rtype.split()
    levels, i = [], 0
    while i < len(toks) and toks[i] in ("*const", "*mut"):
        levels.append("const" if toks[i] == "*const" else "mut")
        i += 1
    base = " ".join(toks[i:])
    kind = RS_SCALAR.get(base, (base, None))[0]
    return tuple(["ptr"] * len(levels) + levels + [kind]) if levels else (kind,)


def strip_c(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def parse_header(text):
    text = strip_c(text)
    consts = {m.group(1): int(m.group(2).strip("()")) for m in re.finditer(r"^#define\s+(MLDSA_[A-Z0-9_]+)\s+(\(?-?\d+\)?)\s*$", text, flags=re.M)}
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    opaque = set(re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", text))
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(1).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            first, *rest = [x.strip() for x in stmt.split(",")]
            mm = re.match(r"^(.*?)(\w+)$", first)
            ftype, fname = mm.group(1).strip(), mm.group(2)
            base = ftype.replace("*", " ").strip()
            fields.append((fname, c_class(ftype)))
            for r in rest:
                fields.append((r.lstrip("* ").strip(), c_class(base + " " + "*" * r.count("*"))))
        structs[m.group(2)] = fields
    body = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    funcs = {}
    for m in re.finditer(r"([A-Za-z_][\w \*]*?)\b(mldsa_\w+)\s*\(([^()]*)\)\s*;", body, flags=re.S):
        args = []
        a = " ".join(m.group(3).split())
        if a and a != "void":
            for piece in a.split(","):
                mm = re.match(r"^(.*?)(\w+)$", piece.strip())
                args.append(c_class(mm.group(1)))
        funcs[m.group(2)] = (c_class(" ".join(m.group(1).split())), args)
    return consts, opaque, structs, funcs


def parse_rust(text):
    text = re.sub(r"//[^\n]*", "", text)
    consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (MLDSA_\w+): c_int = (-?\d+);", text)}
    opaque, structs = set(), {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^\]]*\)\]\s*)?pub struct (\w+) \{(.*?)\}", text, flags=re.S):
        body = m.group(2)
        if "_private" in body:
            opaque.add(m.group(1))
            continue
        structs[m.group(1)] = [(f.group(1).rstrip("_") if f.group(1) in ("in_", "type_") else f.group(1), rs_class(f.group(2).strip()))
                               for f in re.finditer(r"pub (\w+): ([^,]+),", body)]
    ext = re.search(r'extern "C" \{(.*)\}', text, flags=re.S).group(1)
    funcs = {}
    for m in re.finditer(r"pub fn (\w+)\(([^)]*)\)(?:\s*->\s*([^;]+))?;", ext, flags=re.S):
        args = [rs_class(a.split(":", 1)[1].strip()) for a in " ".join(m.group(2).split()).split(", ") if a.strip()]
        funcs[m.group(1)] = (rs_class(m.group(3).strip()) if m.group(3) else ("void",), args)
    return consts, opaque, structs, funcs


def compare(header_text, rust_text):
    """list of differences (empty = the binding matches the header)"""
    hc, ho, hs, hf = parse_header(header_text)
    rc, ro, rs, rf = parse_rust(rust_text)
    diffs = []
    if hc != rc:
        diffs.append(("constants", sorted(set(hc.items()) ^ set(rc.items()))))
    if ho != ro:
        diffs.append(("opaque types", sorted(ho ^ ro)))
    if set(hs) != set(rs):
        diffs.append(("struct names", sorted(set(hs) ^ set(rs))))
    for name in set(hs) & set(rs):
        if hs[name] != rs[name]:
            diffs.append((f"struct {name}", hs[name], rs[name]))
    if set(hf) != set(rf):
        diffs.append(("function names", sorted(set(hf) ^ set(rf))))
    for name in set(hf) & set(rf):
        if hf[name] != rf[name]:
            diffs.append((f"fn {name}", hf[name], rf[name]))
    return diffs


def test_binding_matches_header():
    header, rust = open(HEADER).read(), open(SYS_RS).read()
    assert compare(header, rust) == []
    hc, ho, hs, hf = parse_header(header)
    assert len(hf) >= 88 and {"mldsa_expand_s", "mldsa_to_mont", "mldsa_infinity_norm", "mldsa_pointwise_mont", "mldsa_add_vector_ntt", "mldsa_verify_arith",
                              "mldsa_get_params", "mldsa_get_stats", "mldsa_get_option", "mldsa_device_count", "mldsa_ctx_device", "mldsa_ctx_malloc",
                              "mldsa_memset", "mldsa_profile_enable", "mldsa_profile_report", "mldsa_debug_count_nonzero"} <= set(hf)  # the 16 missing in round 4
    assert set(hs) == {"mldsa_params", "mldsa_stats", "mldsa_verify_slice", "mldsa_sign_slice", "mldsa_keygen_slice", "mldsa_batcher_stats"}
    assert ho == {"mldsa_ctx", "mldsa_group", "mldsa_batcher"}
    # spot checks of the classes themselves (a parser that mapped everything to one class would pass the comparison)
    assert hf["mldsa_ntt"] == (("int",), [("ptr", "mut", "mldsa_ctx"), ("ptr", "const", "i32"), ("ptr", "mut", "i32"), ("usize",), ("ptr", "mut", "void")])
    assert hf["mldsa_last_error"] == (("ptr", "const", "char"), [])
    assert hf["mldsa_ctx_create"][1][1] == ("ptr", "ptr", "mut", "mut", "mldsa_ctx")
    assert hf["mldsa_group_allgather"][1][1] == ("ptr", "ptr", "const", "mut", "u8")       # uint8_t *const *bufs -> *const *mut u8
    assert hf["mldsa_set_option"][1][2] == ("long",) and hf["mldsa_get_option"][0] == ("long",)
    assert [n for n, _ in hs["mldsa_keygen_slice"]] == ["xi", "pk", "sk", "n_keys", "stream"]
    assert hs["mldsa_stats"] == [(n, ("u64",)) for n in ("graphs_captured", "graph_replays", "direct_calls", "workspace_growths", "sign_extra_rounds", "workspace_shrinks")]


def test_generated_file_is_current():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_comparison_notices_a_changed_prototype():
    header, rust = open(HEADER).read(), open(SYS_RS).read()
    mutations = [
        ("int mldsa_ntt(mldsa_ctx *ctx, const int32_t *w, int32_t *w_hat, size_t n_polys, void *stream);",
         "int mldsa_ntt(mldsa_ctx *ctx, const int32_t *w, int32_t *w_hat, size_t n_polys, int flags, void *stream);", "fn mldsa_ntt"),           # arity
        ("int mldsa_to_mont(mldsa_ctx *ctx, const int32_t *in,", "int mldsa_to_mont(mldsa_ctx *ctx, int32_t *in,", "fn mldsa_to_mont"),             # constness
        ("int mldsa_reserve(mldsa_ctx *ctx, int set, int op, size_t n_ops);", "int mldsa_reserve(mldsa_ctx *ctx, int set, int op, int n_ops);", "fn mldsa_reserve"),  # width
        ("int mldsa_device_count(void);", "int mldsa_device_count(void);\nint mldsa_brand_new_call(int x);", "function names"),                    # a new entry point
        ("    size_t n_keys;\n    void *stream;\n} mldsa_keygen_slice;", "    void *stream;\n    size_t n_keys;\n} mldsa_keygen_slice;", "struct mldsa_keygen_slice"),  # field order
        ("uint64_t keys_expanded, key_hits;", "uint64_t keys_expanded, key_hits, evictions;", "struct mldsa_batcher_stats"),                          # a grown struct
        ("#define MLDSA_ERR_AGAIN (-5)", "#define MLDSA_ERR_AGAIN (-6)", "constants"),
    ]
    for old, new, where in mutations:
        assert old in header, old
        diffs = compare(header.replace(old, new), rust)
        assert diffs and any(d[0] == where for d in diffs), (where, diffs)


def test_shim_crates_are_files_and_use_only_declared_functions():
    """rust/fips204-hip: the batch front-end and the single-operation shim as source files; every sys:: function they call is declared"""
    base = os.path.join(ROOT, "rust")
    for rel in ("fips204-hip-sys/Cargo.toml", "fips204-hip-sys/build.rs", "fips204-hip-sys/src/lib.rs", "fips204-hip/Cargo.toml", "fips204-hip/src/lib.rs",
                "fips204-hip/src/batch.rs", "fips204-hip/src/single_op.rs"):
        assert os.path.getsize(os.path.join(base, rel)) > 200, rel
    _, _, _, rf = parse_rust(open(SYS_RS).read())
    consts = parse_rust(open(SYS_RS).read())[0]
    used_f, used_c = set(), set()
    for rel in ("lib.rs", "batch.rs", "single_op.rs"):
        src = open(os.path.join(base, "fips204-hip", "src", rel)).read()
        used_f |= set(re.findall(r"sys::(mldsa_[a-z0-9_]+)\s*\(", src))
        used_c |= set(re.findall(r"sys::(MLDSA_[A-Z0-9_]+)", src))
    assert used_f and used_f <= set(rf), sorted(used_f - set(rf))
    assert used_c <= set(consts), sorted(used_c - set(consts))
    assert {"mldsa_batcher_verify", "mldsa_batcher_sign", "mldsa_batcher_keygen", "mldsa_batcher_forget_key", "mldsa_verify_host", "mldsa_sign_host",
            "mldsa_keygen_host", "mldsa_verify_group"} <= used_f
    # arity of every call site = arity of the declaration
    for rel in ("lib.rs", "batch.rs", "single_op.rs"):
        src = open(os.path.join(base, "fips204-hip", "src", rel)).read()
        for m in re.finditer(r"sys::(mldsa_[a-z0-9_]+)\s*\(", src):
            depth, i, n_args, seen = 1, m.end(), 0, False
            while depth:
                ch = src[i]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                elif ch == "," and depth == 1:
                    n_args += 1
                if depth and not ch.isspace():
                    seen = True
                i += 1
            n_args = n_args + 1 if seen else 0
            assert n_args == len(rf[m.group(1)][1]), (rel, m.group(1), n_args, len(rf[m.group(1)][1]))


def test_integration_md_quotes_the_files_not_a_copy():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "rust/fips204-hip-sys/src/lib.rs" in text and "rust/fips204-hip/src/single_op.rs" in text
    assert text.count("pub fn mldsa_") <= 3   # the declarations live in the generated file; the document shows at most an excerpt


def test_single_op_shim_runs_hashml_dsa_on_the_device_and_drop_never_creates_a_batcher():
    """VERDICT r5 "What's missing" 4 / ADVICE r5: hash_verify / try_hash_sign_* forwarded to the reference's CPU body under a comment that
    claimed the device; Drop for the private key called batcher() (lazy device initialisation + expect inside drop) and every clone's
    drop forgot the key.  rustc is absent: textual guards on the one file."""
    src = open(os.path.join(ROOT, "rust", "fips204-hip", "src", "single_op.rs")).read()
    code = re.sub(r"//[^\n]*", "", src)
    assert "self.inner.hash_verify" not in code and "self.inner.try_hash_sign" not in code
    assert code.count("sys::MLDSA_MODE_PREHASH") == 2        # one verify site, one sign site (both hash-sign entry points share it)
    # the OID of each pre-hash function and the digest length that follows it (src/hashing.rs:317-354; FIPS 204 Algorithm 4 lines 10-22)
    assert "[0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02]" in code
    for name, last, total in (("SHA256", "0x01", 43), ("SHA512", "0x03", 75), ("SHAKE128", "0x0B", 43)):
        m = re.search(r"Ph::%s => \{(.*?)\n        \}" % name, code, flags=re.S)
        assert m and "out[10] = %s;" % last in m.group(1) and m.group(1).rstrip().endswith(str(total)), name
    # the same bytes as the Python mirror's pre-hash (which tests/test_gpu_host_api.py runs against the oracle on the device)
    from fips204_amd import ml_dsa
    for name, last in (("SHA256", 0x01), ("SHA512", 0x03), ("SHAKE128", 0x0B)):
        oid_phm = ml_dsa.hash_message(b"abc", getattr(ml_dsa, "PH_" + name))
        assert oid_phm[:11] == bytes([0x06, 0x09, 0x60, 0x86, 0x48, 0x01, 0x65, 0x03, 0x04, 0x02, last]) and len(oid_phm) == (75 if name == "SHA512" else 43)
    # Drop: only an EXISTING batcher is asked to forget the key, and only by the last owner of the shared wire bytes
    drop = re.search(r"impl Drop for SkWire \{(.*?)\n            \}", code, flags=re.S).group(1)
    assert "existing_batcher()" in drop and "batcher()" not in drop.replace("existing_batcher()", "")
    assert "wire: Arc<SkWire>" in code and "impl Drop for HipPrivateKey" not in code
    cargo = open(os.path.join(ROOT, "rust", "fips204-hip", "Cargo.toml")).read()
    assert re.search(r'^sha2 = ', cargo, flags=re.M) and re.search(r'^sha3 = ', cargo, flags=re.M)
