"""Constants of the measurement: peaks, parameter sets, the reference's published single-core figures."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

Q = 8380417
KECCAK_PEAK_MEASURED_GPERMS = 9.26  # tools/ubench_valu.hip k_keccak at 8 waves/SIMD (profiles/r01_ubench_valu.txt): cross-check only
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured-achievable
# The integer-issue ceiling of the lane-per-state Keccak-f[1600] (csrc/keccak.h), DERIVED from its instruction mix and the
# measured issue cost of each instruction class on this chip (profiles/r01_ubench_valu.txt, cycles per wave64 instruction per
# SIMD at the measured clock): one round = 70 v_bitop3_b32 (chi, theta parities) + 58 v_alignbit_b32 (rotates) + 62 v_xor_b32.
KECCAK_ROUND_MIX = {"v_bitop3_b32": (70, 4.4), "v_alignbit_b32": (58, 4.4), "v_xor_b32": (62, 2.7)}  # (count per round, cycles)
GPU_SIMDS, GPU_CLOCK_GHZ = 256 * 4, 2.4


def keccak_issue_ceiling():
    """G permutations/s if every SIMD issued nothing but Keccak rounds, with the arithmetic spelled out"""
    cyc_round = sum(n * c for n, c in KECCAK_ROUND_MIX.values())
    cyc_perm = 24 * cyc_round            # per wave = per 64 states
    peak = GPU_SIMDS * GPU_CLOCK_GHZ * 64 / cyc_perm
    return peak, {
        "instruction_mix_per_round": {k: {"count": n, "issue_cycles_per_wave64_instruction": c} for k, (n, c) in KECCAK_ROUND_MIX.items()},
        "issue_costs_source": "profiles/r01_ubench_valu.txt (tools/ubench_valu.hip, column cyc/instr@clk; v_bitop3_b32 issues like v_bfi_b32 / v_and_or_b32)",
        "cycles_per_round_per_wave": cyc_round, "rounds": 24, "cycles_per_permutation_per_wave": cyc_perm, "states_per_wave": 64,
        "simds": GPU_SIMDS, "clock_GHz": GPU_CLOCK_GHZ,
        "formula": "simds * clock_GHz * states_per_wave / cycles_per_permutation_per_wave",
        "G_permutations_per_s": peak,
        "measured_pure_keccak_kernel_G_per_s": KECCAK_PEAK_MEASURED_GPERMS,
    }


KECCAK_PEAK_GPERMS, KECCAK_PEAK_DERIVATION = keccak_issue_ceiling()
SETS = {44: dict(k=4, l=4, gamma1=1 << 17, tau=39), 65: dict(k=6, l=5, gamma1=1 << 19, tau=49),
        87: dict(k=8, l=7, gamma1=1 << 19, tau=60)}


# the reference's own published single-core figures (benches/README.md:16-26; i7-7700K @ 4.2 GHz, Rust 1.81,
# RUSTFLAGS="-C target-cpu=native" cargo bench), printed beside the CPU baseline measured here
REFERENCE_PUBLISHED = {
    "source": "/root/reference/benches/README.md:16-26 (Intel i7-7700K @ 4.20 GHz, one core, Oct 2024)",
    "keygen_us": {44: 104.89, 65: 194.80, 87: 290.24},
    "sign_us": {44: 226.32, 65: 352.89, 87: 385.05},
    "verify_us": {44: 21.016, 65: 27.996, 87: 36.468},
    "note": ("published figures of another machine, quoted for context only.  They are not mutually consistent by operation count: "
             "keygen (about 190 Keccak-f incl. ExpandA) is listed at 194.8 us, verify (159 Keccak-f incl. the same ExpandA, "
             "ml_dsa.rs:406) at 28.0 us.  The oracle timed here spends 318 us per keygen at 2.1 GHz against the published 194.8 us "
             "at 4.2-4.5 GHz."),
}
