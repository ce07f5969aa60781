#!/usr/bin/env python3
"""Vector-issue model of render_kernel<false>: the kernel's own dynamic opcode histogram (exact per-block execution
counts from tools/bbprof.py x the instructions of every block) priced with the per-opcode issue costs tools/valu_rate.hip measures on
gfx950, against the SIMD cycles the launch really had.

  python tools/issue_model.py <bbprof outdir> [--valu-rate profiles/r04_valu_rate.txt] [--measured-cycles-per-valu X]

<bbprof outdir> = gpurun_out/<tag> of tools/bbprof.sh: device.s, blocks.json, profile.json.
Prints markdown: the histogram by cost class, the predicted cycles per vector instruction (mix average) and, given the measured cycles per
vector instruction per SIMD of the full launch (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs / SQ_INSTS_VALU), the share of the SIMD's time the
vector pipe is predicted to be busy.
"""
import argparse
import collections
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bbprof  # noqa: E402

# Issue cost classes (cycles per wave64 instruction per SIMD with >= 4 resident waves; profiles/r04_valu_rate.txt).  An opcode that the
# micro-benchmark does not cover is priced by the class of its nearest relative and listed under "assumed".
FULL = 2.15    # v_add/sub/mul/fmac_f32, v_mov, v_add/sub_u32, v_and/or/xor/not, v_ashrrev
FMA = 2.45     # v_fma_f32, v_bitop3 (VOP3, three register operands): 2.60 at 4 waves, 2.24 at 8
HALF = 4.15    # compares, selects (VOP3 form), min / max / med3, conversions, floor / ceil / rndne, v_lshlrev, v_lshl_add, ffbl / ffbh, v_mul_lo, div_scale / fmas / fixup
TRANS = 8.10   # v_rcp_f32 (v_sqrt, v_rsq)
CND_VOP2 = 2.2      # v_cndmask_b32_e32 next to other work (tools/valu_rate.hip: cnd_vcc_add_mix)
CND_VOP2_B2B = 15.8  # v_cndmask_b32_e32 right behind another one (cnd_vcc_by_vcmp)

MEASURED = {  # opcode prefix -> class (covered by tools/valu_rate.hip)
    "v_add_f32": FULL, "v_sub_f32": FULL, "v_subrev_f32": FULL, "v_mul_f32": FULL, "v_fmac_f32": FULL, "v_mov_b32": FULL, "v_add_u32": FULL, "v_sub_u32": FULL,
    "v_subrev_u32": FULL, "v_and_b32": FULL, "v_or_b32": FULL, "v_not_b32": FULL, "v_ashrrev_i32": FULL, "v_xor_b32": FULL, "v_lshrrev_b32": FULL,
    "v_fma_f32": FMA, "v_bitop3_b32": FMA,
    "v_lshlrev_b32": HALF, "v_lshl_add_u32": HALF, "v_min_f32": HALF, "v_max_f32": HALF, "v_min_i32": HALF, "v_max_i32": HALF, "v_min_u32": HALF, "v_max_u32": HALF, "v_med3_f32": HALF,
    "v_cvt_": HALF, "v_floor_f32": HALF, "v_ceil_f32": HALF, "v_rndne_f32": HALF, "v_trunc_f32": HALF, "v_ffbl_b32": HALF, "v_ffbh_u32": HALF, "v_mul_lo_u32": HALF,
    "v_div_scale_f32": HALF, "v_div_fmas_f32": HALF, "v_div_fixup_f32": HALF, "v_cmp_": HALF, "v_cmpx_": HALF, "v_cndmask_b32_e64": HALF, "v_addc_co_u32": HALF,
    "v_rcp_f32": TRANS, "v_sqrt_f32": TRANS, "v_rsq_f32": TRANS,
}
ASSUMED_HALF = ("v_bfe_", "v_bfrev_", "v_mul_u32_u24", "v_mul_i32_i24", "v_mad_", "v_add3_", "v_or3_", "v_and_or_", "v_lshl_or_", "v_add_lshl_", "v_readfirstlane", "v_readlane", "v_mbcnt", "v_alignbit", "v_perm", "v_add_co_", "v_sub_co_", "v_mul_hi_")


def price(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op == "v_cndmask_b32_e64":
        return HALF, "measured"
    for k, v in MEASURED.items():
        if base.startswith(k) or op.startswith(k):
            return v, "measured"
    for k in ASSUMED_HALF:
        if base.startswith(k):
            return HALF, "assumed"
    return HALF, "assumed"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--measured-cycles-per-valu", type=float, default=None, help="SIMD cycles of the launch / vector instructions executed (per SIMD)")
    a = ap.parse_args()
    lines = open(os.path.join(a.outdir, "device.s")).read().split("\n")
    blocks = bbprof.parse_blocks(lines)
    prof = json.load(open(os.path.join(a.outdir, "profile.json")))
    execs = {int(k): v for k, v in prof["executions"].items()}
    dyn = collections.Counter()
    cnd_b2b = 0.0
    cnd_e32 = 0.0
    salu = branch = waits = vmem = lds = 0.0
    for b, blk in enumerate(blocks):
        e = execs.get(b, 0.0)
        if not e:
            continue
        prev = None
        for op in blk["ins"]:
            kind = bbprof.classify(op)
            if kind == "valu":
                dyn[op] += e
                if op == "v_cndmask_b32_e32":
                    cnd_e32 += e
                    if prev == "v_cndmask_b32_e32":
                        cnd_b2b += e
            elif kind == "salu":
                salu += e
            elif kind == "branch":
                branch += e
            elif kind == "wait":
                waits += e
            elif kind == "lds":
                lds += e
            elif kind == "vmem":
                vmem += e
            prev = op
    total = sum(dyn.values())
    measured_valu = prof["base"].get("SQ_INSTS_VALU", 0.0)
    by_class = collections.defaultdict(float)
    cycles = 0.0
    assumed = collections.Counter()
    for op, n in dyn.items():
        if op == "v_cndmask_b32_e32":
            continue
        c, how = price(op)
        by_class[c] += n
        cycles += c * n
        if how == "assumed":
            assumed[op] += n
    cyc_lo = cycles + cnd_e32 * CND_VOP2                                   # every VOP2 select at its interleaved price
    cyc_hi = cycles + (cnd_e32 - cnd_b2b) * CND_VOP2 + cnd_b2b * CND_VOP2_B2B  # ... and the back-to-back ones at theirs
    print(f"# Vector-issue model of `render_kernel<false>` ({os.path.basename(os.path.normpath(a.outdir))})\n")
    print(f"Dynamic vector instructions of the profiled dispatch: {total:.5g} (hardware `SQ_INSTS_VALU` {measured_valu:.5g}: {'exact' if abs(total - measured_valu) < 1e-6 * max(1.0, measured_valu) else 'MISMATCH'}); "
          f"scalar {salu:.4g}, branches {branch:.4g}, `s_waitcnt` / `s_nop` {waits:.4g}, vector memory {vmem:.4g}, LDS {lds:.4g}.\n")
    print("| issue class (cycles per wave64 instruction per SIMD) | dynamic instructions | share | cycles |")
    print("|---|---|---|---|")
    names = {FULL: "full rate", FMA: "`v_fma_f32` / `v_bitop3`", HALF: "half rate", TRANS: "`v_rcp_f32` (quarter rate)"}
    for c in sorted(by_class):
        print(f"| {names[c]} ({c}) | {by_class[c]:.4g} | {100 * by_class[c] / total:.1f} % | {by_class[c] * c:.4g} |")
    print(f"| `v_cndmask_b32_e32` not behind another one ({CND_VOP2}) | {cnd_e32 - cnd_b2b:.4g} | {100 * (cnd_e32 - cnd_b2b) / total:.1f} % | {(cnd_e32 - cnd_b2b) * CND_VOP2:.4g} |")
    print(f"| `v_cndmask_b32_e32` right behind another one ({CND_VOP2} ... {CND_VOP2_B2B}) | {cnd_b2b:.4g} | {100 * cnd_b2b / total:.1f} % | {cnd_b2b * CND_VOP2:.4g} ... {cnd_b2b * CND_VOP2_B2B:.4g} |")
    print(f"\nPredicted vector-pipe cycles per vector instruction (mix average): **{cyc_lo / total:.3f}** (back-to-back VOP2 selects at the interleaved price) ... **{cyc_hi / total:.3f}** (at the back-to-back price).")
    if assumed:
        print("\nPriced by assumption (half rate), not measured: " + ", ".join(f"`{op}` {n / total * 100:.2f} %" for op, n in assumed.most_common(12)) + ".")
    print("\nTop opcodes: " + ", ".join(f"`{op}` {100 * n / total:.1f} %" for op, n in dyn.most_common(16)) + ".")
    if a.measured_cycles_per_valu:
        m = a.measured_cycles_per_valu
        print(f"\nMeasured: {m:.3f} SIMD cycles per vector instruction over the whole launch -> the vector pipe is predicted busy **{100 * cyc_lo / total / m:.0f} % ... {100 * cyc_hi / total / m:.0f} %** of the time.")
    print(f"\nScalar side of the same dispatch: {salu + branch:.4g} scalar + branch instructions = {(salu + branch) / total:.3f} per vector instruction; at the 4 cycles per scalar instruction and SIMD of `salu_and` that is "
          f"{4.0 * (salu + branch) / total:.2f} cycles per vector instruction of a pipe of its own (it overlaps with other waves' vector instructions: `salu_valu_mix`).")


if __name__ == "__main__":
    main()
