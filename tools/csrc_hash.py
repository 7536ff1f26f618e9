#!/usr/bin/env python3
"""Content hash of the library's sources (fips204_amd/csrc/*.hip, *.h, *.cpp, Makefile and include/mldsa_hip.h): what ties a profile
under profiles/ to the code it was taken on (profiles/rNN_MANIFEST.json; tests/test_profiles_manifest_cpu.py).  Works without git (the
GPU box receives a snapshot without .git/)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_files(root=ROOT):
    d = os.path.join(root, "fips204_amd", "csrc")
    names = sorted(n for n in os.listdir(d) if n.endswith((".hip", ".h", ".cpp")) or n == "Makefile")
    return [os.path.join("fips204_amd", "csrc", n) for n in names] + [os.path.join("include", "mldsa_hip.h")]


def csrc_hash(root=ROOT):
    h = hashlib.sha256()
    for rel in source_files(root):
        data = open(os.path.join(root, rel), "rb").read()
        h.update(rel.encode() + b"\0" + str(len(data)).encode() + b"\0" + data)
    return h.hexdigest()


def lib_sha256(root=ROOT):
    p = os.path.join(root, "fips204_amd", "csrc", "libmldsa_hip.so")
    return hashlib.sha256(open(p, "rb").read()).hexdigest() if os.path.exists(p) else None


if __name__ == "__main__":
    print(csrc_hash() if len(sys.argv) < 2 or sys.argv[1] != "lib" else lib_sha256())
