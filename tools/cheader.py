"""A small parser for include/mldsa_hip.h: constants, structs, prototypes.

Used by tools/gen_rust_sys.py (which writes rust/fips204-hip-sys/src/lib.rs from the header) -- and NOT by
tests/test_rust_binding_cpu.py, which parses the header on its own so that a bug here cannot hide a drift."""
import re


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def split_decl(decl):
    """'const uint8_t *rho' -> ('const uint8_t *', 'rho'); 'uint8_t *const *bufs' -> ('uint8_t *const *', 'bufs')"""
    decl = " ".join(decl.split())
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)$", decl)
    ctype, name = m.group(1).strip(), m.group(2)
    ctype = re.sub(r"\s*\*\s*", " *", ctype).strip()
    ctype = ctype.replace("* *", "**").replace("*const", "* const")
    return " ".join(ctype.split()), name


def parse(path):
    raw = open(path).read()
    text = strip_comments(raw)
    consts = [(m.group(1), m.group(2).strip()) for m in re.finditer(r"^#define\s+(MLDSA_[A-Z0-9_]+)\s+(\(?-?\d+\)?)\s*$", text, flags=re.M)]
    consts = [(n, int(v.strip("()"))) for n, v in consts if n != "MLDSA_HIP_H"]
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = re.sub(r'extern\s+"C"\s*\{', "", text)
    opaque, structs, funcs = [], [], []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
        opaque.append(m.group(2))
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(1).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            # 'const uint8_t *rho, *tr' / 'int set, k, l' / 'unsigned long long a, b'
            parts = [p.strip() for p in stmt.split(",")]
            base_t, first = split_decl(parts[0])
            base = base_t.rstrip("* ").strip()
            fields.append((base_t, first))
            for p in parts[1:]:
                stars = len(p) - len(p.lstrip("* "))
                nstar = p.count("*")
                fields.append(((base + " " + "*" * nstar).strip(), p.lstrip("* ").strip()))
        structs.append((m.group(2), fields))
    body = re.sub(r"typedef\s+struct\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)
    body = re.sub(r"typedef\s+struct\s+\w+\s+\w+\s*;", "", body)
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(mldsa_\w+)\s*\(([^()]*)\)\s*;", body, flags=re.S):
        ret = " ".join(m.group(1).split())
        ret = re.sub(r"\s*\*\s*", " *", ret).strip()
        args = []
        arg_text = " ".join(m.group(3).split())
        if arg_text and arg_text != "void":
            for a in arg_text.split(","):
                args.append(split_decl(a))
        funcs.append((m.group(2), ret, args))
    return dict(consts=consts, opaque=opaque, structs=structs, funcs=funcs)
