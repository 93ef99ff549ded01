"""VALU issue rate of the chip, one opcode at a time and in the mix of K6's Smith-Waterman (lh_diag_valu_rate, csrc/k_valu_rate.h).
    python tools/valu_rate.py [--iters N] > profiles/r06_valu_rate.log
Columns: G wave-instructions/s over the whole chip (HIP events), cycles per wave64 instruction per SIMD (event time x median shader clock x SIMDs / instructions),
the median cycles per instruction seen by ONE wave (s_memtime around its loop), the median and the lowest shader clock of the launch
(s_memtime / s_memrealtime), and how the waves were spread (SIMDs seen, fewest-most waves on one SIMD)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from lariat_amd import capi

OPS = [
    (0, "v_add_u32"), (1, "v_fma_f32"), (17, "v_add_f32"), (18, "v_xor_b32"), (8, "v_max_u32"), (9, "v_max3_u32"), (14, "v_add3_u32"), (10, "v_and_or_b32"), (21, "v_sub_u32 clamp"),
    (12, "v_max_u16"), (15, "v_sad_u8"),
    (2, "v_pk_max_u16"), (11, "v_pk_min_u16"), (19, "v_pk_max_i16"), (3, "v_pk_sub_u16 clamp"), (13, "v_pk_sub_u16 clamp (sgpr)"), (4, "v_pk_add_u16"), (22, "v_pk_mad_u16"), (16, "v_pk_add_f16"),
    (5, "v_perm_b32"), (7, "v_lshl_or_b32"), (6, "v_mov_b32 dpp row_shr:1 (+s_nop 1)"), (20, "v_max_u32 dpp row_shr:1 (+s_nop 1)"),
    (23, "v_max_f32"), (24, "v_max3_f32"), (39, "v_med3_f32"), (38, "v_sub_f32 clamp"), (54, "v_fmac_f32"), (55, "v_cvt_f32_ubyte0"), (48, "v_max_f16"),
    (25, "v_min_u32"), (26, "v_max_i32"), (42, "v_min_i32"), (47, "v_min3_u32"), (49, "v_sub_u32"), (27, "v_cndmask_b32 (vcc)"), (43, "v_cmp_eq_u32 -> vcc"), (28, "v_mov_b32"),
    (29, "v_and_b32"), (30, "v_or_b32"), (40, "v_or3_b32"), (44, "v_bfi_b32"), (31, "v_lshlrev_b32"), (46, "v_lshl_add_u32"), (37, "v_bfe_u32"), (35, "v_mul_u32_u24"), (36, "v_mad_u32_u24"),
    (32, "v_add_u16"), (33, "v_sub_u16 clamp"), (34, "v_max_i16"), (53, "v_max_u16 (e64, sgpr)"), (45, "v_mad_u16"), (41, "v_max_u16 sdwa WORD_1"),
    (51, "v_pk_fma_f32"), (52, "s_nop 0"), (56, "alternating v_pk_max_u16 / v_max_u16 (32 + 32)"), (57, "v_pk_max_u16 + 3 full-rate (16 + 48)"),
    (58, "v_add_u32 / v_xor_b32 / v_sub_u32 / v_and_b32 rotating (all full-rate, 32-bit)"), (59, "v_max_u16 / v_sub_u16 clamp / v_add_u16 rotating (all full-rate, 16-bit)"),
    (60, "v_max_u16 / v_add_u32 alternating (full-rate, 16- and 32-bit)"), (61, "2 v_pk_max_u16 then 2 v_add_u32 (32 + 32)"), (50, "k_resc_sw column mix (1 perm 1 add 6 subs 7 max 1 lshl_or)"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=4096)
    ap.add_argument("--waves", default="1,2,4,8")
    a = ap.parse_args()
    lib = capi.load_library()
    waves = [int(w) for w in a.waves.split(",")]
    print("# lh_diag_valu_rate: 64 x %d wave64 instructions per wave; blocks of 4 waves, CUs x W blocks" % a.iters)
    print("%-58s %2s %9s %9s %9s %7s %7s %6s %5s" % ("op (8 independent chains)", "W", "G instr/s", "cyc/SIMD", "cyc/wave", "MHz", "minMHz", "SIMDs", "w/SIMD"))
    for op, name in OPS:
        for w in waves:
            r = lib.diag_valu_rate(op, w, a.iters)
            print("%-58s %2d %9.1f %9.2f %9.2f %7.0f %7.0f %6d %2d-%-2d" % (name, w, r["ginstr_per_s"], r["cycles_per_instr_simd"], r["cycles_per_instr_wave"], r["mhz"], r["min_mhz"],
                                                                           r["simds"], r["min_waves_simd"], r["max_waves_simd"]))
        sys.stdout.flush()
    print()
    print("%-58s %2s %9s %9s" % ("op (ONE dependent chain)", "W", "cyc/SIMD", "cyc/wave"))
    for op, name in OPS:
        if op in (50, 51, 52, 43) or op >= 56:
            continue
        for w in (1, 4):
            r = lib.diag_valu_rate(100 + op, w, a.iters)
            print("%-58s %2d %9.2f %9.2f" % (name, w, r["cycles_per_instr_simd"], r["cycles_per_instr_wave"]))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
