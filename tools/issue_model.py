#!/usr/bin/env python3
"""Instruction-issue model of render_kernel<false>: the kernel's own dynamic opcode histogram (exact per-block execution
counts from tools/bbprof.py x the instructions of every block) priced with the issue costs tools/valu_rate.hip measures on gfx950
(profiles/r04_valu_rate.txt), against the SIMD cycles the launch really had.

  python tools/issue_model.py <bbprof outdir> [--cycles-per-instruction X]

<bbprof outdir> = gpurun_out/<tag> of tools/bbprof.sh: device.s, blocks.json, profile.json.
--cycles-per-instruction: measured SIMD cycles per executed instruction of the full launch,
  = (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs / (SQ_INSTS_VALU + SALU + BRANCH + LDS + VMEM), from the counter summaries of the same build.

The prices (cycles of ONE SIMD's issue per wave64 instruction, four or more resident waves, instructions of different kinds interleaved -- the
`mix_*` rows, which is how a real kernel presents them; the single-kind rows measure the throughput of one pipe, e.g. 4.1 for a stream of nothing
but compares, and do not add up in a mix):
  vector, any kind (add / mul / fma / compare / select / min / max / convert / shift / div_scale ...)   2.25
  v_rcp_f32 (v_sqrt, v_rsq)                                                                           8.1 (stream of them) ... 13.2 (one per four, mix_rcp)
  v_pk_*_f32                                                                                          4.1 ... 17 (mix_pk_fma); the build has none (-fno-slp-vectorize)
  scalar ALU, branch: a pipe of the CU (one instruction per cycle for its four SIMDs = 4.0 per SIMD when every SIMD issues them, salu_and);
      between vector instructions they overlap only in part: 1.0 (the price of an s_nop slot) ... 2.25 (mix_salu2: two scalar + two vector cost 4 x 2.25)
  s_nop / s_waitcnt (satisfied)                                                                       0.9
  LDS / vector memory instruction                                                                     2.3 (mix_lds)
"""
import argparse
import collections
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bbprof  # noqa: E402

VALU = 2.25
TRANS = (8.1, 13.2)
SCALAR = (1.0, 2.25)
WAIT = 0.9
MEM = 2.3
# single-kind throughput of the vector pipes (profiles/r04_valu_rate.txt, 4 waves per SIMD) -- reported beside the issue model: no pipe is the limit
PIPE_FULL, PIPE_FMA, PIPE_HALF = 2.15, 2.45, 4.15
FULL_RATE = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fmac_f32", "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_not_b32",
             "v_ashrrev_i32", "v_xor_b32", "v_lshrrev_b32")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("outdir")
    ap.add_argument("--cycles-per-instruction", type=float, default=None)
    a = ap.parse_args()
    lines = open(os.path.join(a.outdir, "device.s")).read().split("\n")
    blocks = bbprof.parse_blocks(lines)
    prof = json.load(open(os.path.join(a.outdir, "profile.json")))
    execs = {int(k): v for k, v in prof["executions"].items()}
    n = collections.Counter()
    ops = collections.Counter()
    for b, blk in enumerate(blocks):
        e = execs.get(b, 0.0)
        if not e:
            continue
        for op in blk["ins"]:
            kind = bbprof.classify(op)
            base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
            if kind == "valu":
                ops[base] += e
                if base in ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32"):
                    n["trans"] += e
                elif base.startswith("v_pk_"):
                    n["packed"] += e
                else:
                    n["valu"] += e
                    n["pipe_full" if base.startswith(FULL_RATE) else ("pipe_fma" if base in ("v_fma_f32", "v_bitop3_b32") else "pipe_half")] += e
            elif kind in ("salu", "branch"):
                n[kind] += e
            elif kind == "wait":
                n["wait"] += e
            else:
                n["mem"] += e
    vec = n["valu"] + n["trans"] + n["packed"]
    total = vec + n["salu"] + n["branch"] + n["mem"]  # (waits are not counted by SQ_INSTS_*: left out of the per-instruction figures, priced below)
    hw = prof["base"].get("SQ_INSTS_VALU", 0.0)
    lo = n["valu"] * VALU + n["trans"] * TRANS[0] + n["packed"] * 4.1 + (n["salu"] + n["branch"]) * SCALAR[0] + n["wait"] * WAIT + n["mem"] * MEM
    hi = n["valu"] * VALU + n["trans"] * TRANS[1] + n["packed"] * 17.0 + (n["salu"] + n["branch"]) * SCALAR[1] + n["wait"] * WAIT + n["mem"] * MEM
    print(f"# Instruction-issue model of `render_kernel<false>` ({os.path.basename(os.path.normpath(a.outdir))})\n")
    print(f"Dynamic instructions of the profiled dispatch (exact: executions of every basic block x its instructions; vector total {vec:.5g} against the hardware's `SQ_INSTS_VALU` {hw:.5g}: "
          f"{'identical' if abs(vec - hw) < 1e-6 * max(1.0, hw) else ('within %.2f %%' % (100 * abs(vec - hw) / max(1.0, hw)) if abs(vec - hw) < 5e-3 * hw else 'MISMATCH')}):\n")
    print("| kind | dynamic instructions | per vector instruction | issue price (cycles of one SIMD) | cycles |")
    print("|---|---|---|---|---|")
    rows = [("vector, not transcendental", n["valu"], f"{VALU}", f"{n['valu'] * VALU:.4g}"),
            ("`v_rcp_f32`", n["trans"], f"{TRANS[0]} ... {TRANS[1]}", f"{n['trans'] * TRANS[0]:.4g} ... {n['trans'] * TRANS[1]:.4g}"),
            ("scalar ALU", n["salu"], f"{SCALAR[0]} ... {SCALAR[1]}", f"{n['salu'] * SCALAR[0]:.4g} ... {n['salu'] * SCALAR[1]:.4g}"),
            ("branch", n["branch"], f"{SCALAR[0]} ... {SCALAR[1]}", f"{n['branch'] * SCALAR[0]:.4g} ... {n['branch'] * SCALAR[1]:.4g}"),
            ("`s_waitcnt` / `s_nop`", n["wait"], f"{WAIT}", f"{n['wait'] * WAIT:.4g}"),
            ("LDS + vector memory", n["mem"], f"{MEM}", f"{n['mem'] * MEM:.4g}")]
    for name, cnt, price, cyc in rows:
        print(f"| {name} | {cnt:.4g} | {cnt / vec:.3f} | {price} | {cyc} |")
    print(f"| **all** | {total + n['wait']:.4g} | {(total + n['wait']) / vec:.3f} | | **{lo:.4g} ... {hi:.4g}** |")
    print(f"\nPredicted issue cycles per executed instruction (vector + scalar + branch + memory, the instructions `SQ_INSTS_*` count): **{lo / total:.3f} ... {hi / total:.3f}**.")
    if a.cycles_per_instruction:
        m = a.cycles_per_instruction
        print(f"Measured over the whole 512-frame launch: **{m:.3f}** SIMD cycles per executed instruction -> the SIMDs' issue is predicted busy **{100 * lo / total / m:.0f} % ... {100 * hi / total / m:.0f} %** of the launch "
              "(lower end: every scalar instruction hidden behind another wave's vector instruction as well as an `s_nop` is; upper end: scalar instructions at the price `mix_salu2` measures at this occupancy).")
    pipe = n["pipe_full"] * PIPE_FULL + n["pipe_fma"] * PIPE_FMA + n["pipe_half"] * PIPE_HALF + n["trans"] * TRANS[0]
    print(f"\nFor comparison, the vector PIPES (single-kind throughputs; they work side by side, only the busiest matters): full-rate pipe {n['pipe_full'] * PIPE_FULL + n['pipe_fma'] * PIPE_FMA:.4g} cycles "
          f"({n['pipe_full'] + n['pipe_fma']:.4g} instructions), half-rate pipe {n['pipe_half'] * PIPE_HALF + n['trans'] * TRANS[0]:.4g} cycles ({n['pipe_half']:.4g} instructions + the reciprocals) "
          f"-- {100 * max(n['pipe_full'] * PIPE_FULL + n['pipe_fma'] * PIPE_FMA, n['pipe_half'] * PIPE_HALF + n['trans'] * TRANS[0]) / (vec * VALU):.0f} % of the vector issue time above: no pipe is the limit, the issue is.  "
          f"(Priced additively, as if one pipe took every instruction at its single-kind rate: {pipe:.4g}.)")
    print(f"Scalar unit of the CU (one instruction per cycle for four SIMDs): {4.0 * (n['salu'] + n['branch']):.4g} cycles per SIMD = {100 * 4.0 * (n['salu'] + n['branch']) / (total * (a.cycles_per_instruction or 2.56)):.0f} % of the launch.")
    print("\nTop vector opcodes: " + ", ".join(f"`{op}` {100 * c / vec:.1f} %" for op, c in ops.most_common(14)) + ".")


if __name__ == "__main__":
    main()
