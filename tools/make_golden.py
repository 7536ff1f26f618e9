"""Regenerates tests/golden/ from the reference's own test DATA (run in the build container, where
/root/reference is mounted).  Only data is copied: the three public NIST ACVP-Server JSON files the
reference keeps under tests/nist_vectors/ (gzip, re-serialised compactly) and the hex literals of
tests/messages.rs:18-20 and tests/integration.rs:64-73.  No reference source text is stored."""
import gzip
import json
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = "tests/golden"

for n in ("keyGen", "sigGen", "sigVer"):
    v = json.load(open(f"{REF}/tests/nist_vectors/ML-DSA-{n}-FIPS204/internalProjection.json"))
    with gzip.GzipFile(f"{OUT}/acvp_{n}.json.gz", "wb", mtime=0) as f:
        f.write(json.dumps(v, separators=(",", ":")).encode())

msg = open(f"{REF}/tests/messages.rs").read()
hexes = re.findall(r'hex::decode\("([0-9a-fA-F]+)"\)', msg)
assert len(hexes) == 3
out = {"messages_rs": {"msg_ascii": "asdf", "ctx_hex": "", "rng": "ChaCha8Rng::seed_from_u64(123)",
                       "sk": hexes[0], "sig": hexes[1], "pk": hexes[2]}}
integ = open(f"{REF}/tests/integration.rs").read()
seg = integ[integ.index("fn bad_sig"):integ.index("fn test_44_no_verif")]
hx = re.findall(r'hex::decode\("([0-9a-fA-F]+)"\)', seg)
assert [len(h) // 2 for h in hx] == [32, 2560, 1312, 2420, 2420]
out["integration_bad_sig"] = {"msg": hx[0], "sk": hx[1], "pk": hx[2], "good_sig": hx[3], "bad_sig": hx[4]}
with gzip.GzipFile(f"{OUT}/reference_hex_vectors.json.gz", "wb", mtime=0) as f:
    f.write(json.dumps(out).encode())
print("tests/golden regenerated")
