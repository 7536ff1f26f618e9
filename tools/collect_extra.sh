#!/bin/bash
# One more measurement on the library the round's profiles were taken on, added to the manifest afterwards:
#     gpurun -- 'tools/collect_extra.sh r06 ab_small_sign_spec.txt "<what>" <command...>'     (stdout of the command -> the file)
#     python tools/finish_profiles.py r06 --extra      (back in the container: copies gpurun_out/extra_r06/* into profiles/ and appends them to
#                                                        the manifest -- refused unless the sources' hash and the library's sha256 are the manifest's)
set -u
R=$1; F=$2; WHAT=$3; shift 3
OUT=$PWD/gpurun_out/extra_$R
mkdir -p "$OUT"
python3 - > "$OUT/BUILD.json" <<PY
import json, sys
sys.path.insert(0, "tools")
import csrc_hash
print(json.dumps({"lib_sha256": csrc_hash.lib_sha256(), "csrc_hash": csrc_hash.csrc_hash()}))
PY
"$@" > "$OUT/$F" 2>> "$OUT/stderr.log"
python3 -c 'import json,sys; print(json.dumps({"file": sys.argv[1], "command": sys.argv[2]}))' "$F" "$WHAT" >> "$OUT/MANIFEST.jsonl"
cat "$OUT/$F"
