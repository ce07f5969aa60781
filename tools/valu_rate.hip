// valu_rate.hip -- diagnostic micro-benchmark (not part of the product): what one SIMD of gfx950 pays per wave64
// instruction, by instruction kind and by resident waves per SIMD, in REAL shader cycles.
//
// Round 4 rewrite.  What was wrong with the round-3 tool and what this one does instead:
//   * cycles were wall time x a nominal 2.4 GHz: now every wave stamps s_memtime (shader clock) and s_memrealtime
//     (constant 100 MHz) around its loop; cycles are the stamps' difference, the clock is their ratio;
//   * "waves per SIMD" was assumed from the grid size with one-wave workgroups: now ONE workgroup of 256 x W threads
//     per CU (a workgroup's waves are dealt 0 -> 2 -> 1 -> 3 over the SIMDs; the whole LDS is requested so that a CU
//     takes one workgroup, or half of it for two), the waves start together behind a barrier, and a census of
//     HW_REG_HW_ID / HW_REG_XCC_ID says how many waves every SIMD really held (printed when it is not W everywhere);
//   * the `cndmask` row read a VCC that nothing in the kernel had written (23 cycles): now the select is measured with its
//     mask in VCC written by v_cmp, in an SGPR pair written by v_cmp, in an SGPR pair written by s_and_b64 / s_mov_b64,
//     and in the never-written VCC of the old row -- the forms the render kernel uses and the one that made the outlier;
//   * the scalar rows printed 0.01 (their bodies named no operand, so the assembler text was the same for every KIND and
//     the compiler merged the loops away): every body now names an operand.
//
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate [filter]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct WaveRec {
	unsigned long long t0, t1, real;
	unsigned hwid, xcc;
};

// Bodies: 16 instructions on 8 independent destination registers (%0..%7), sources %8 / %9 (VGPR), repeated 4 x per loop trip.
// Clobbered scalars: vcc, s[20:27].
#define L8(a, b) a "%0" b "\n" a "%1" b "\n" a "%2" b "\n" a "%3" b "\n" a "%4" b "\n" a "%5" b "\n" a "%6" b "\n" a "%7" b "\n"
#define D8(a, b) a "%0" b "%0\n" a "%1" b "%1\n" a "%2" b "%2\n" a "%3" b "%3\n" a "%4" b "%4\n" a "%5" b "%5\n" a "%6" b "%6\n" a "%7" b "%7\n"
#define T2(x) x x

#define BODY_LIST(X)                                                                                                            \
	X(fma, "", T2(D8("v_fma_f32 ", ", %8, %9, ")))                                                                              \
	X(fmac, "", T2(L8("v_fmac_f32 ", ", %8, %9")))                                                                              \
	X(mul, "", T2(D8("v_mul_f32 ", ", %8, ")))                                                                                  \
	X(add, "", T2(D8("v_add_f32 ", ", %8, ")))                                                                                  \
	X(sub, "", T2(D8("v_sub_f32 ", ", %8, ")))                                                                                  \
	X(mov, "", T2(L8("v_mov_b32 ", ", %8")))                                                                                    \
	X(add_u32, "", T2(D8("v_add_u32 ", ", %8, ")))                                                                              \
	X(sub_u32, "", T2(D8("v_sub_u32 ", ", %8, ")))                                                                              \
	X(and_b32, "", T2(D8("v_and_b32 ", ", %8, ")))                                                                              \
	X(or_b32, "", T2(D8("v_or_b32 ", ", %8, ")))                                                                                \
	X(not_b32, "", T2(D8("v_not_b32 ", ", ")))                                                                                  \
	X(lshlrev, "", T2(D8("v_lshlrev_b32 ", ", 1, ")))                                                                           \
	X(ashrrev, "", T2(D8("v_ashrrev_i32 ", ", 1, ")))                                                                           \
	X(lshl_add, "", T2(D8("v_lshl_add_u32 ", ", %8, 3, ")))                                                                     \
	X(bitop3, "", T2(D8("v_bitop3_b32 ", ", %8, %9, ") ))                                                                       \
	X(min_f32, "", T2(D8("v_min_f32 ", ", %8, ")))                                                                              \
	X(max_f32, "", T2(D8("v_max_f32 ", ", %8, ")))                                                                              \
	X(min_i32, "", T2(D8("v_min_i32 ", ", %8, ")))                                                                              \
	X(med3, "", T2(D8("v_med3_f32 ", ", %8, %9, ")))                                                                            \
	X(cvt_f32_i32, "", T2(D8("v_cvt_f32_i32 ", ", ")))                                                                          \
	X(cvt_i32_f32, "", T2(D8("v_cvt_i32_f32 ", ", ")))                                                                          \
	X(cvt_flr, "", T2(D8("v_cvt_flr_i32_f32 ", ", ")))                                                                          \
	X(floor, "", T2(D8("v_floor_f32 ", ", ")))                                                                                  \
	X(rndne, "", T2(D8("v_rndne_f32 ", ", ")))                                                                                  \
	X(ffbl, "", T2(D8("v_ffbl_b32 ", ", ")))                                                                                    \
	X(mul_lo, "", T2(D8("v_mul_lo_u32 ", ", %8, ")))                                                                            \
	X(rcp, "", T2(D8("v_rcp_f32 ", ", ")))                                                                                      \
	X(div_scale, "", T2(L8("v_div_scale_f32 ", ", vcc, %8, %9, %8")))                                                           \
	X(div_fmas, "v_cmp_lt_f32 vcc, %8, %9\n", T2(D8("v_div_fmas_f32 ", ", %8, %9, ")))                                          \
	X(div_fixup, "", T2(L8("v_div_fixup_f32 ", ", %0, %8, %9")))                                                                \
	X(cmp_vcc, "", T2(L8("v_cmp_lt_f32 vcc, %8, ", "")))                                                                        \
	X(cmp_sgpr, "", "v_cmp_lt_f32 s[20:21], %8, %0\n v_cmp_lt_f32 s[22:23], %8, %1\n v_cmp_lt_f32 s[24:25], %8, %2\n v_cmp_lt_f32 s[26:27], %8, %3\n" \
	                "v_cmp_lt_f32 s[20:21], %8, %4\n v_cmp_lt_f32 s[22:23], %8, %5\n v_cmp_lt_f32 s[24:25], %8, %6\n v_cmp_lt_f32 s[26:27], %8, %7\n" \
	                "v_cmp_lt_i32 s[20:21], %8, %0\n v_cmp_lt_i32 s[22:23], %8, %1\n v_cmp_lt_i32 s[24:25], %8, %2\n v_cmp_lt_i32 s[26:27], %8, %3\n" \
	                "v_cmp_lt_i32 s[20:21], %8, %4\n v_cmp_lt_i32 s[22:23], %8, %5\n v_cmp_lt_i32 s[24:25], %8, %6\n v_cmp_lt_i32 s[26:27], %8, %7\n") \
	/* the select, by where its lane mask comes from */                                                                         \
	X(cnd_vcc_by_vcmp, "v_cmp_lt_f32 vcc, %8, %9\n", T2(L8("v_cndmask_b32 ", ", %8, %9, vcc")))                                       \
	X(cnd_sgpr_by_vcmp, "v_cmp_lt_f32 s[20:21], %8, %9\n", T2(L8("v_cndmask_b32_e64 ", ", %8, %9, s[20:21]")))                  \
	X(cnd_sgpr_by_salu, "s_and_b64 s[20:21], exec, s[22:23]\n", T2(L8("v_cndmask_b32_e64 ", ", %8, %9, s[20:21]")))             \
	X(cnd_vcc_by_salu, "s_and_b64 vcc, exec, s[22:23]\n", T2(L8("v_cndmask_b32 ", ", %8, %9, vcc")))                            \
	X(cnd_vcc_stale, "", T2(L8("v_cndmask_b32 ", ", %8, %9, vcc")))                                                             \
	X(cnd_sgpr_stale, "", T2(L8("v_cndmask_b32_e64 ", ", %8, %9, s[24:25]")))                                                   \
	X(cmp_cnd3, "", "v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n" \
	                "v_cmp_lt_f32 vcc, %8, %3\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n" \
	                "v_cmp_lt_f32 vcc, %8, %6\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n v_cndmask_b32 %0, %8, %9, vcc\n" \
	                "v_cmp_lt_f32 vcc, %8, %1\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n") \
	X(cmp_add_cnd, "", "v_cmp_lt_f32 vcc, %8, %0\n v_add_f32 %4, %8, %4\n v_cndmask_b32 %0, %8, %9, vcc\n v_add_f32 %5, %8, %5\n" \
	                   "v_cmp_lt_f32 vcc, %8, %1\n v_add_f32 %6, %8, %6\n v_cndmask_b32 %1, %8, %9, vcc\n v_add_f32 %7, %8, %7\n" \
	                   "v_cmp_lt_f32 vcc, %8, %2\n v_add_f32 %4, %8, %4\n v_cndmask_b32 %2, %8, %9, vcc\n v_add_f32 %5, %8, %5\n" \
	                   "v_cmp_lt_f32 vcc, %8, %3\n v_add_f32 %6, %8, %6\n v_cndmask_b32 %3, %8, %9, vcc\n v_add_f32 %7, %8, %7\n") \
	X(cnd_e64_vcc, "v_cmp_lt_f32 vcc, %8, %9\n", T2(L8("v_cndmask_b32_e64 ", ", %8, %9, vcc")))                                \
	X(cnd_vcc_dst_src, "v_cmp_lt_f32 vcc, %8, %9\n", T2(D8("v_cndmask_b32 ", ", %8, ") ))   /* (vcc implied) */                \
	X(cnd_vcc_add_mix, "v_cmp_lt_f32 vcc, %8, %9\n", "v_cndmask_b32 %0, %8, %9, vcc\n v_add_f32 %4, %8, %4\n v_cndmask_b32 %1, %8, %9, vcc\n v_add_f32 %5, %8, %5\n" \
	                   "v_cndmask_b32 %2, %8, %9, vcc\n v_add_f32 %6, %8, %6\n v_cndmask_b32 %3, %8, %9, vcc\n v_add_f32 %7, %8, %7\n" \
	                   "v_cndmask_b32 %0, %8, %9, vcc\n v_add_f32 %4, %8, %4\n v_cndmask_b32 %1, %8, %9, vcc\n v_add_f32 %5, %8, %5\n" \
	                   "v_cndmask_b32 %2, %8, %9, vcc\n v_add_f32 %6, %8, %6\n v_cndmask_b32 %3, %8, %9, vcc\n v_add_f32 %7, %8, %7\n") \
	X(addc_vcc, "v_cmp_lt_f32 vcc, %8, %9\n", T2(L8("v_addc_co_u32 ", ", vcc, %8, %9, vcc")))                                        \
	/* a fresh mask for every select: compare + select pairs, and s_and + select pairs (what `a & b ? x : y` compiles to) */    \
	X(cmp_cnd_pairs, "", "v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %9, vcc\n v_cmp_lt_f32 vcc, %8, %1\n v_cndmask_b32 %1, %8, %9, vcc\n" \
	                     "v_cmp_lt_f32 vcc, %8, %2\n v_cndmask_b32 %2, %8, %9, vcc\n v_cmp_lt_f32 vcc, %8, %3\n v_cndmask_b32 %3, %8, %9, vcc\n" \
	                     "v_cmp_lt_f32 vcc, %8, %4\n v_cndmask_b32 %4, %8, %9, vcc\n v_cmp_lt_f32 vcc, %8, %5\n v_cndmask_b32 %5, %8, %9, vcc\n" \
	                     "v_cmp_lt_f32 vcc, %8, %6\n v_cndmask_b32 %6, %8, %9, vcc\n v_cmp_lt_f32 vcc, %8, %7\n v_cndmask_b32 %7, %8, %9, vcc\n") \
	X(sand_cnd_pairs, "", "s_and_b64 s[20:21], exec, s[22:23]\n v_cndmask_b32_e64 %0, %8, %9, s[20:21]\n s_or_b64 s[24:25], exec, s[22:23]\n v_cndmask_b32_e64 %1, %8, %9, s[24:25]\n" \
	                      "s_and_b64 s[20:21], exec, s[22:23]\n v_cndmask_b32_e64 %2, %8, %9, s[20:21]\n s_or_b64 s[24:25], exec, s[22:23]\n v_cndmask_b32_e64 %3, %8, %9, s[24:25]\n" \
	                      "s_and_b64 s[20:21], exec, s[22:23]\n v_cndmask_b32_e64 %4, %8, %9, s[20:21]\n s_or_b64 s[24:25], exec, s[22:23]\n v_cndmask_b32_e64 %5, %8, %9, s[24:25]\n" \
	                      "s_and_b64 s[20:21], exec, s[22:23]\n v_cndmask_b32_e64 %6, %8, %9, s[20:21]\n s_or_b64 s[24:25], exec, s[22:23]\n v_cndmask_b32_e64 %7, %8, %9, s[24:25]\n") \
	/* round 4, second question: is a compare / select through VCC (VOP2 / VOPC encoding) cheaper than the VOP3 form with an SGPR pair when other    \
	   work sits between them?  Four instructions per group, one compare or select in each */                                                       \
	X(mix_cmp_e64, "", T2("v_cmp_lt_f32 s[20:21], %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"                    \
	                      "v_cmp_lt_f32 s[22:23], %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))                   \
	X(mix_cmp_e32, "", T2("v_cmp_lt_f32 vcc, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"                         \
	                      "v_cmp_lt_f32 vcc, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))                        \
	X(mix_cmp_e32_smov, "", "v_cmp_lt_f32 vcc, %8, %0\n s_mov_b64 s[20:21], vcc\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n" \
	                        "v_cmp_lt_f32 vcc, %8, %4\n s_mov_b64 s[22:23], vcc\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n" \
	                        "v_cmp_lt_f32 vcc, %8, %0\n s_mov_b64 s[24:25], vcc\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n" \
	                        "v_cmp_lt_f32 vcc, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")                       \
	X(mix_cnd_e64, "", T2("v_cndmask_b32_e64 %0, %8, %9, s[22:23]\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"           \
	                      "v_cndmask_b32_e64 %4, %8, %9, s[24:25]\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))          \
	X(mix_cnd_e32_smov, "", "s_mov_b64 vcc, s[22:23]\n v_cndmask_b32 %0, %8, %9, vcc\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n" \
	                        "s_mov_b64 vcc, s[24:25]\n v_cndmask_b32 %4, %8, %9, vcc\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n" \
	                        "s_mov_b64 vcc, s[22:23]\n v_cndmask_b32 %0, %8, %9, vcc\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n" \
	                        "v_cndmask_b32 %4, %8, %9, vcc\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")                  \
	X(mix_min, "", T2("v_min_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"                               \
	                  "v_min_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))                                \
	X(mix_lshl, "", T2("v_lshlrev_b32 %0, 1, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"                            \
	                   "v_lshlrev_b32 %4, 1, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))                            \
	X(mix_cmp_cnd_e64, "", T2("v_cmp_lt_f32 s[20:21], %8, %0\n v_add_f32 %1, %8, %1\n v_cndmask_b32_e64 %2, %8, %9, s[20:21]\n v_add_f32 %3, %8, %3\n" \
	                          "v_cmp_lt_f32 s[22:23], %8, %4\n v_add_f32 %5, %8, %5\n v_cndmask_b32_e64 %6, %8, %9, s[22:23]\n v_add_f32 %7, %8, %7\n")) \
	X(mix_cmp_cnd_e32, "", T2("v_cmp_lt_f32 vcc, %8, %0\n v_add_f32 %1, %8, %1\n v_cndmask_b32 %2, %8, %9, vcc\n v_add_f32 %3, %8, %3\n"           \
	                          "v_cmp_lt_f32 vcc, %8, %4\n v_add_f32 %5, %8, %5\n v_cndmask_b32 %6, %8, %9, vcc\n v_add_f32 %7, %8, %7\n"))         \
	X(mix_rcp, "", T2("v_rcp_f32 %0, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"                                    \
	                  "v_rcp_f32 %4, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))                                     \
	X(mix_div_scale, "", T2("v_div_scale_f32 %0, vcc, %8, %9, %8\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"            \
	                        "v_div_fixup_f32 %4, %4, %8, %9\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))               \
	X(mix_snop, "", T2("s_nop 0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"                                          \
	                   "s_nop 0\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))                                           \
	X(mix_waitcnt, "", T2("s_waitcnt vmcnt(0)\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"                            \
	                      "s_waitcnt lgkmcnt(0)\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))                           \
	X(mix_salu2, "", T2("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[24:25], s[24:25], s[26:27]\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n" \
	                    "s_andn2_b64 s[22:23], s[22:23], s[24:25]\n s_xor_b64 s[26:27], s[26:27], s[20:21]\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n")) \
	/* round 5: does a scalar instruction cost an issue slot beside vector work, or only when the CU's one scalar unit is saturated?        \
	   mix_salu2 is 50 % scalar: four SIMDs x 2 of 4 instructions ask the scalar unit for ~0.9 instructions per cycle -- its limit, whatever issue costs.       \
	   The kernel's own ratio is ~1 scalar / branch per 2.6 vector: mix_salu1 (1 of 4) and mix_salu3of8 keep the scalar unit at <= 60 %. */                 \
	X(mix_salu1, "", T2("s_and_b64 s[20:21], s[20:21], s[22:23]\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"                 \
	                    "s_or_b64 s[24:25], s[24:25], s[26:27]\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))                \
	X(mix_salu3of8, "", T2("s_and_b64 s[20:21], s[20:21], s[22:23]\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n s_or_b64 s[24:25], s[24:25], s[26:27]\n" \
	                       "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n s_andn2_b64 s[22:23], s[22:23], s[24:25]\n v_add_f32 %7, %8, %7\n"))           \
	X(mix_salu1_dep, "", T2("v_cmp_lt_f32 vcc, %8, %0\n s_and_b64 s[20:21], vcc, s[22:23]\n v_cndmask_b32_e64 %2, %8, %2, s[20:21]\n v_add_f32 %3, %8, %3\n" \
	                        "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))                             \
	X(mix_branch1, "", T2("v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n s_cbranch_execz 9f\n"                                   \
	                      "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n s_cbranch_execz 9f\n") "9:\n")                           \
	X(mix_saveexec, "", T2("v_cmp_lt_f32 vcc, %8, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %2, %8, %2\n s_or_b64 exec, exec, s[20:21]\n"    \
	                       "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n"))                         \
	X(mix_pk_fma, "", "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_pk_fma_f32 %[p0], %[p2], %[p2], %[p0]\n v_add_f32 %7, %8, %7\n" \
	                  "v_add_f32 %0, %8, %0\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_pk_fma_f32 %[p1], %[p2], %[p2], %[p1]\n v_add_f32 %7, %8, %7\n") \
	X(mix_lds, "", T2("ds_read_b32 %0, %[a]\n v_add_f32 %1, %8, %1\n v_add_f32 %2, %8, %2\n v_add_f32 %3, %8, %3\n"                               \
	                  "ds_read_b32 %4, %[a]\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n") "s_waitcnt lgkmcnt(0)\n")    \
	X(mix_add_only, "", T2(D8("v_add_f32 ", ", %8, ")))                                                                            \
	/* scalar side */                                                                                                           \
	X(salu_and, "", "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[22:23], s[24:25]\n s_and_b64 s[24:25], s[24:25], s[26:27]\n s_or_b64 s[26:27], s[26:27], s[20:21]\n" \
	                "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[22:23], s[24:25]\n s_and_b64 s[24:25], s[24:25], s[26:27]\n s_or_b64 s[26:27], s[26:27], s[20:21]\n" \
	                "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[22:23], s[24:25]\n s_and_b64 s[24:25], s[24:25], s[26:27]\n s_or_b64 s[26:27], s[26:27], s[20:21]\n" \
	                "s_and_b64 s[20:21], s[20:21], s[22:23]\n s_or_b64 s[22:23], s[22:23], s[24:25]\n s_and_b64 s[24:25], s[24:25], s[26:27]\n v_mov_b32 %0, %8\n") \
	X(salu_valu_mix, "", "s_and_b64 s[20:21], s[20:21], s[22:23]\n v_add_f32 %0, %8, %0\n s_or_b64 s[22:23], s[22:23], s[24:25]\n v_add_f32 %1, %8, %1\n" \
	                     "s_and_b64 s[24:25], s[24:25], s[26:27]\n v_add_f32 %2, %8, %2\n s_or_b64 s[26:27], s[26:27], s[20:21]\n v_add_f32 %3, %8, %3\n" \
	                     "s_and_b64 s[20:21], s[20:21], s[22:23]\n v_add_f32 %4, %8, %4\n s_or_b64 s[22:23], s[22:23], s[24:25]\n v_add_f32 %5, %8, %5\n" \
	                     "s_and_b64 s[24:25], s[24:25], s[26:27]\n v_add_f32 %6, %8, %6\n s_or_b64 s[26:27], s[26:27], s[20:21]\n v_add_f32 %7, %8, %7\n") \
	X(saveexec_region, "", "v_cmp_lt_f32 vcc, %8, %0\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %0, %8, %0\n s_or_b64 exec, exec, s[20:21]\n" \
	                       "v_cmp_lt_f32 vcc, %8, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %1, %8, %1\n s_or_b64 exec, exec, s[20:21]\n" \
	                       "v_cmp_lt_f32 vcc, %8, %2\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %2, %8, %2\n s_or_b64 exec, exec, s[20:21]\n" \
	                       "v_cmp_lt_f32 vcc, %8, %3\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f32 %3, %8, %3\n s_or_b64 exec, exec, s[20:21]\n") \
	X(branch_not_taken, "", "v_cmp_lt_f32 vcc, %8, %0\n s_cbranch_vccz 1f\n v_add_f32 %0, %8, %0\n1:\n v_cmp_lt_f32 vcc, %8, %1\n s_cbranch_vccz 2f\n v_add_f32 %1, %8, %1\n2:\n" \
	                        "v_cmp_lt_f32 vcc, %8, %2\n s_cbranch_vccz 3f\n v_add_f32 %2, %8, %2\n3:\n v_cmp_lt_f32 vcc, %8, %3\n s_cbranch_vccz 4f\n v_add_f32 %3, %8, %3\n4:\n" \
	                        "v_add_f32 %4, %8, %4\n v_add_f32 %5, %8, %5\n v_add_f32 %6, %8, %6\n v_add_f32 %7, %8, %7\n") \
	X(snop, "", "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n v_mov_b32 %0, %8\n") \
	/* dependent chains (latency of one wave's own stream): every instruction reads the previous one's result */               \
	X(dep_fma, "", "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n" \
	               "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n" \
	               "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n" \
	               "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n") \
	X(dep_rcp, "", "v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n" \
	               "v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n v_rcp_f32 %0, %0\n") \
	X(dep_cmp_cnd, "", "v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %0, vcc\n v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %0, vcc\n" \
	                   "v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %0, vcc\n v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %0, vcc\n" \
	                   "v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %0, vcc\n v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %0, vcc\n" \
	                   "v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %0, vcc\n v_cmp_lt_f32 vcc, %8, %0\n v_cndmask_b32 %0, %8, %0, vcc\n")

enum Kind {
#define X(name, pre, text) K_##name,
	BODY_LIST(X)
#undef X
	K_count
};

__device__ __forceinline__ unsigned long long realtime()
{
	unsigned long long t;
	asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
	return t;
}
__device__ __forceinline__ unsigned long long shadertime()
{
	unsigned long long t;
	asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
	return t;
}

// `active` < 64: only that many lanes of every wave run the loop (the rest wait at the end)
template <int KIND>
__global__ __launch_bounds__(1024) void k(WaveRec *recs, float *out, float seed, int iters, int active, int spread = 0)
{
	extern __shared__ unsigned ldsPad[];
	const int lane = threadIdx.x & 63;
	float a0 = seed + lane, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
	float m = 1.0000001f + seed * 1e-9f, c = 1e-9f + seed;
	double pd0 = a0, pd1 = a1, pd2 = m; // register pairs for the packed form
	const unsigned ldsAddr = (threadIdx.x & 1023u) * 4u;
	if (seed == 12345.f) { ldsPad[threadIdx.x] = 1u; }
	// the "written by the scalar unit long ago" masks: an alternating lane pattern
	asm volatile("s_mov_b64 s[22:23], 0x55555555\n s_mov_b64 s[24:25], 0x33333333\n s_mov_b64 s[26:27], -1\n s_mov_b64 s[20:21], 0" ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
	__syncthreads();
	unsigned long long t0 = 0, r0 = 0;
	const bool on = spread ? ((lane * active) >> 6) != (((lane - 1) * active) >> 6) || lane == 0 : lane < active; // spread: `active` lanes at even distances
	if (on) {
		t0 = shadertime();
		r0 = realtime();
		for (int i = 0; i < iters; i++) {
#define X(name, pre, text)                                                                                                        \
	if (KIND == K_##name) {                                                                                                       \
		asm volatile(pre text text text text : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
		             : "v"(m), "v"(c), [p2] "v"(pd2), [a] "v"(ldsAddr), [p0] "v"(pd0), [p1] "v"(pd1) /* (p0 / p1 are written by the packed test: timing only) */ \
		                          : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");                                      \
	}
			BODY_LIST(X)
#undef X
		}
		const unsigned long long t1 = shadertime(), r1 = realtime();
		if (lane == 0) {
			unsigned hwid, xcc;
			asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
			WaveRec r;
			r.t0 = t0;
			r.t1 = t1;
			r.real = r1 - r0;
			r.hwid = hwid;
			r.xcc = xcc;
			recs[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
		}
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(pd0 + pd1);
}

static int instructionsPerTrip(const char *pre, const char *text)
{
	auto count = [](const char *s) { int n = 0; for (const char *p = s; *p; p++) { n += *p == '\n'; } return n; };
	int labels = 0;
	for (const char *p = text; *p; p++) { labels += (p[0] == ':' && p[1] == '\n'); }
	return count(pre) + 4 * (count(text) - labels);
}

struct Result {
	double cyclesPerInstSimd, mhz;
	int minWaves, maxWaves;
};

template <int KIND>
Result run(const char *pre, const char *text, WaveRec *dRecs, float *dOut, int cus, int wavesPerSimd, int active, int spread = 0)
{
	// workgroups: W <= 4: one of 256 W threads per CU; 6: two of 768; 8: two of 1024
	const int perCu = wavesPerSimd <= 4 ? 1 : 2;
	const int threads = 256 * wavesPerSimd / perCu;
	const size_t ldsBytes = perCu == 1 ? 160 * 1024 - 512 : 80 * 1024 - 512;
	const int blocks = cus * perCu;
	const int iters = 4000;
	(void)hipFuncSetAttribute((const void *)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
	const int waves = blocks * threads / 64;
	(void)hipMemset(dRecs, 0, sizeof(WaveRec) * (size_t)waves);
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), ldsBytes, 0, dRecs, dOut, 1.0f, 200, active, spread); // warm-up
	hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), ldsBytes, 0, dRecs, dOut, 1.0f, iters, active, spread);
	if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return { 0, 0, 0, 0 }; }
	std::vector<WaveRec> recs(waves);
	(void)hipMemcpy(recs.data(), dRecs, sizeof(WaveRec) * (size_t)waves, hipMemcpyDeviceToHost);
	// per SIMD: the span from its first wave's start to its last wave's end (the waves of a SIMD do not share it evenly: the oldest
	// is served first, so one wave's own loop time says little); reported: the median SIMD's span / instructions of all its waves
	struct Simd { int waves = 0; unsigned long long first = ~0ull, last = 0; };
	std::map<unsigned, Simd> perSimd;
	std::vector<double> cyc, mhz;
	for (const WaveRec &r : recs) {
		Simd &sd = perSimd[(r.xcc & 0xF) << 16 | (r.hwid & 0xFF30)]; // se, sh, cu, simd (pipe and wave slot masked out)
		sd.waves++;
		sd.first = std::min(sd.first, r.t0);
		sd.last = std::max(sd.last, r.t1);
		mhz.push_back(r.real ? (double)(r.t1 - r.t0) / (double)r.real * 100.0 : 0.0);
	}
	int lo = 1 << 30, hi = 0;
	for (auto &kv : perSimd) {
		lo = std::min(lo, kv.second.waves);
		hi = std::max(hi, kv.second.waves);
		cyc.push_back((double)(kv.second.last - kv.second.first) / kv.second.waves);
	}
	std::sort(cyc.begin(), cyc.end());
	std::sort(mhz.begin(), mhz.end());
	const double inst = (double)iters * instructionsPerTrip(pre, text);
	(void)wavesPerSimd;
	return { cyc[cyc.size() / 2] / inst, mhz[mhz.size() / 2], lo, hi };
}

int main(int argc, char **argv)
{
	const char *filter = argc > 1 ? argv[1] : "";
	hipDeviceProp_t p;
	(void)hipGetDeviceProperties(&p, 0);
	const int cus = p.multiProcessorCount;
	WaveRec *dRecs;
	float *dOut;
	(void)hipMalloc(&dRecs, sizeof(WaveRec) * (size_t)cus * 32);
	(void)hipMalloc(&dOut, sizeof(float) * (size_t)cus * 2048);
	printf("%d CUs.  REAL shader cycles (s_memtime) per wave64 instruction per SIMD = median wave's loop cycles / instructions / waves on its SIMD; (clock MHz = s_memtime / s_memrealtime).\n", cus);
	printf("One 256 W-thread workgroup per CU (W <= 4; two of 128 W for 6 / 8), waves released together by a barrier; census of HW_ID: waves found per SIMD [min..max] printed when not W.\n");
	printf("%-18s %-16s %-16s %-16s %-16s %-16s %-16s\n", "instruction", "1 wave/SIMD", "2", "3", "4", "6", "8");
#define X(name, pre, text)                                                                                   \
	if (strstr(#name, filter)) {                                                                             \
		printf("%-18s", #name);                                                                              \
		for (int w : { 1, 2, 3, 4, 6, 8 }) {                                                                 \
			const Result r = run<K_##name>(pre, text, dRecs, dOut, cus, w, 64);                              \
			char census[32] = "";                                                                            \
			if (r.minWaves != w || r.maxWaves != w) { snprintf(census, sizeof census, "[%d..%d]", r.minWaves, r.maxWaves); } \
			printf(" %5.2f (%4.0f)%-4s", r.cyclesPerInstSimd, r.mhz, census);                                \
		}                                                                                                    \
		printf("\n");                                                                                        \
		fflush(stdout);                                                                                      \
	}
	BODY_LIST(X)
#undef X
	if (strstr("lanes", filter) || !*filter) {
		printf("\nactive lanes per wave, cycles per instruction per SIMD (clock): rows = active lanes, the low ones / spread evenly over the wave\n%-22s", "lanes, instruction");
		for (int w : { 1, 2, 4 }) { printf(" %d wave(s)/SIMD     ", w); }
		printf("\n");
		for (int spread : { 0, 1 })
		for (int active : { 64, 48, 32, 24, 20, 17, 16, 15, 14, 12, 10, 9, 8, 4, 1 }) {
			char label[64];
			snprintf(label, sizeof label, "%2d %s fma", active, spread ? "spread" : "low");
			printf("%-22s", label);
			for (int w : { 1, 2, 4 }) {
				const Result r = run<K_fma>("", T2(D8("v_fma_f32 ", ", %8, %9, ")), dRecs, dOut, cus, w, active, spread);
				printf(" %5.2f (%4.0f)      ", r.cyclesPerInstSimd, r.mhz);
			}
			printf("\n");
		}
		for (int active : { 64, 16, 12, 8, 4 }) {
			char label[64];
			snprintf(label, sizeof label, "%2d low add / cmp_e64 mix", active);
			printf("%-22s", label);
			for (int w : { 1, 2, 4 }) {
				const Result r = run<K_mix_cmp_e64>("", "x\nx\nx\nx\nx\nx\nx\nx\nx\nx\nx\nx\nx\nx\nx\nx\n", dRecs, dOut, cus, w, active, 0);
				printf(" %5.2f (%4.0f)      ", r.cyclesPerInstSimd, r.mhz);
			}
			printf("\n");
		}
	}
	return 0;
}
