#!/usr/bin/env python3
"""After tools/collect_profiles.sh rNN has run on the GPU box: copy gpurun_out/final_rNN/* into profiles/rNN_* and write
profiles/rNN_MANIFEST.json -- per file the command that made it, and for all of them the commit, the sha256 of libmldsa_hip.so and the
hash of the sources they were taken on (tools/csrc_hash.py).    python tools/finish_profiles.py r05"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import csrc_hash  # noqa: E402


def extra(r):
    """files measured later on the SAME library (tools/collect_extra.sh): appended to the manifest, refused if the library differs"""
    src = os.path.join(ROOT, "gpurun_out", f"extra_{r}")
    build = json.load(open(os.path.join(src, "BUILD.json")))
    mp = os.path.join(ROOT, "profiles", f"{r}_MANIFEST.json")
    man = json.load(open(mp))
    if build["csrc_hash"] != man["csrc_hash"] or build["lib_sha256"] != man["lib_sha256"]:
        sys.exit(f"extra files were measured on another library (sources {build['csrc_hash'][:12]} / lib {build['lib_sha256'][:12]}, manifest "
                 f"{man['csrc_hash'][:12]} / {man['lib_sha256'][:12]}): not added")
    for e in (json.loads(ln) for ln in open(os.path.join(src, "MANIFEST.jsonl")) if ln.strip()):
        p = os.path.join(src, e["file"])
        if not os.path.exists(p) or os.path.getsize(p) == 0:
            print("missing or empty:", e["file"])
            continue
        dst = f"{r}_{e['file']}"
        shutil.copyfile(p, os.path.join(ROOT, "profiles", dst))
        man["files"][dst] = {"command": e["command"] + "  (tools/collect_extra.sh: a later run on the same library)", "bytes": os.path.getsize(p)}
        print("added", dst)
    json.dump(man, open(mp, "w"), indent=1, sort_keys=True)


def main():
    r = sys.argv[1] if len(sys.argv) > 1 else "r06"
    if "--extra" in sys.argv:
        return extra(r)
    src = os.path.join(ROOT, "gpurun_out", f"final_{r}")
    build = json.load(open(os.path.join(src, "BUILD.json")))
    entries = [json.loads(ln) for ln in open(os.path.join(src, "MANIFEST.jsonl")) if ln.strip()]
    files = {}
    for e in entries:
        p = os.path.join(src, e["file"])
        if not os.path.exists(p) or os.path.getsize(p) == 0:
            print("missing or empty:", e["file"])
            continue
        dst = f"{r}_{e['file']}"
        shutil.copyfile(p, os.path.join(ROOT, "profiles", dst))
        files[dst] = {"command": e["command"], "bytes": os.path.getsize(p)}
    man = {"round": r, "head": build["head"], "lib_sha256": build["lib_sha256"], "csrc_hash": build["csrc_hash"], "hipcc_version": build.get("hipcc_version"),
           "note": "every file below was produced by tools/collect_profiles.sh on one MI355X box from the library with this sha256, built from the "
                   "sources with this hash (tools/csrc_hash.py) at this commit", "files": files}
    json.dump(man, open(os.path.join(ROOT, "profiles", f"{r}_MANIFEST.json"), "w"), indent=1, sort_keys=True)
    now = csrc_hash.csrc_hash()
    print(f"{len(files)} files; sources then {build['csrc_hash'][:12]} now {now[:12]}: {'CURRENT' if now == build['csrc_hash'] else 'STALE (csrc changed since)'}")


if __name__ == "__main__":
    main()
