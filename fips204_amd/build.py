"""Builds fips204_amd/csrc/libmldsa_hip.so (hipcc, --offload-arch=gfx950) in-tree."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libmldsa_hip.so")


def build(force=False, jobs=8):
    args = ["make", "-C", CSRC, f"-j{jobs}"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    if not os.path.exists(LIB):
        raise RuntimeError(f"build did not produce {LIB}")
    return LIB


if __name__ == "__main__":
    print(build())
