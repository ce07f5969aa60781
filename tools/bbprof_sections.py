#!/usr/bin/env python3
"""Section summary of a basic-block profile (tools/bbprof.py): dynamic instructions per wave-step by part of ExecuteRay.

usage: python tools/bbprof_sections.py <dir with blocks.json, profile.json, device.s>   (prints markdown)

A block belongs to the section of the highest line of trace_ray's own body (cvx_kernels.h) that one of its instructions carries;
blocks that only hold inlined helper code or compiler-made control flow inherit the section of the block before them in code order.
Numbers are per wave-step of the busier of the two instances of trace_ray (iteration direction +1 / -1)."""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = sys.argv[1]
meta = json.load(open(os.path.join(d, "blocks.json")))
prof = json.load(open(os.path.join(d, "profile.json")))
ex = {int(k): v for k, v in prof["executions"].items()}
lines = open(os.path.join(d, "device.s")).read().split("\n")
src = open(os.path.join(ROOT, "cpuvox_amd", "csrc", "cvx_kernels.h")).read().split("\n")


def line_of(pattern, after=0):
    return next(i + 1 for i, l in enumerate(src) if i + 1 > after and re.search(pattern, l))


# `.loc <file> <line>`: only lines of cvx_kernels.h itself count (inlined library code, e.g. __ffs, carries line numbers of other files)
kfile = next((m.group(1) for l in lines for m in [re.match(r'\s*\.file\s+(\d+)\s+.*cvx_kernels\.h"', l)] if m), None)
body = line_of(r"void trace_ray\(")
marks = [
    ("prologue (DDA setup, first column)", body),
    ("drawColumn: Q corners", line_of(r"auto drawColumn = ")),
    ("frustum clip (:295-422)", line_of(r"if \(curDistLast > 2\.0f && frustumDirMaxWorld == ")),
    ("run selection (:424-475)", line_of(r"CVX_END\(2\);")),
    ("side of a run: projection, horizon (:478-517)", line_of(r"const float portionBottom = ")),
    ("side pixels (:519-533)", line_of(r"// pixel loop :519-533")),
    ("top / bottom face: projection, horizon (:544-593)", line_of(r"// top / bottom of the run")),
    ("top / bottom pixels (:595-603)", line_of(r"// :595-603")),
    ("drawColumn: exit", line_of(r"return COUNT \|\| \(!windowClosed")),
    ("prologue (DDA setup, first column)", line_of(r"// column 0: LOD check")),
    ("column step: DDA step, LOD check, record address + fetch, cull (:237-281,613)", line_of(r"auto columnStep = ")),
    ("kernel epilogue (skybox pass)", line_of(r"^template <bool COUNT>", line_of(r"auto columnStep = "))),
]
marks.sort(key=lambda t: t[1])


def section_of_line(n):
    name = None
    for nm, start in marks:
        if n >= start:
            name = nm
    return name


def block_range(i):
    a = meta[i]["line"]
    z = meta[i + 1]["line"] if i + 1 < len(meta) else len(lines)
    return a, z


# the two instances: blocks holding the two look-ahead loads
heads = []
for i, b in enumerate(meta):
    a, z = block_range(i)
    if sum(1 for l in lines[a:z] if "global_load_dwordx4" in l) >= 2:
        heads.append((i, ex.get(b["block"], 0)))
# instance boundary: the kernel emits instance A completely, then instance B; split at the first block of the second half of `heads`
heads = [h for h in heads if h[1] > 1000]
half = len(heads) // 2
split = heads[half][0] - 3
inst = [(0, split), (split, len(meta))]
steps = [sum(e for i, e in heads if lo <= i < hi) for lo, hi in inst]
pick = 0 if steps[0] >= steps[1] else 1
lo, hi = inst[pick]
S = steps[pick]

sec = collections.defaultdict(lambda: collections.Counter())
current = "prologue (DDA setup, first column)"
for i in range(lo, hi):
    b = meta[i]
    a, z = block_range(i)
    own = []
    loc = None
    ins = []
    for l in lines[a:z]:
        s = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            loc = int(m.group(2)) if kfile is None or m.group(1) == kfile else None
            continue
        if not s or s[0] in ".;" or s.endswith(":"):
            continue
        ins.append(s.split()[0])
        if loc and loc >= body and loc < marks[-1][1] + 200:
            own.append(loc)
    if own:
        current = section_of_line(max(own))
    e = ex.get(b["block"], 0)
    for op in ins:
        k = ("valu" if op.startswith("v_") else "branch" if op.startswith("s_cbranch") or op == "s_branch" else "wait" if op.startswith("s_waitcnt") or op == "s_nop"
             else "salu" if op.startswith("s_") else "mem")
        sec[current][k] += e
        if op.startswith("v_mov"):
            sec[current]["v_mov"] += e
    sec[current]["blocks"] += 1

order = []
for nm, _ in marks:
    if nm not in order:
        order.append(nm)
tot = collections.Counter()
print(f"wave-steps of the profiled instance: {S:.0f} (32 frames, iteration direction {'+1' if pick else '-1'} instance = the busier one)\n")
print("| section (reference lines) | VALU | of which v_mov | SALU | branch | memory | waits | total / step | share |")
print("|---|---|---|---|---|---|---|---|---|")
grand = sum(sum(v for k, v in sec[nm].items() if k in ("valu", "salu", "branch", "mem")) for nm in order)
for nm in order:
    c = sec[nm]
    t = c["valu"] + c["salu"] + c["branch"] + c["mem"]
    for k in ("valu", "salu", "branch", "mem", "wait", "v_mov"):
        tot[k] += c[k]
    print(f"| {nm} | {c['valu'] / S:.1f} | {c['v_mov'] / S:.1f} | {c['salu'] / S:.1f} | {c['branch'] / S:.1f} | {c['mem'] / S:.1f} | {c['wait'] / S:.1f} | {t / S:.1f} | {100 * t / grand:.1f} % |")
t = tot["valu"] + tot["salu"] + tot["branch"] + tot["mem"]
print(f"| **all** | {tot['valu'] / S:.1f} | {tot['v_mov'] / S:.1f} | {tot['salu'] / S:.1f} | {tot['branch'] / S:.1f} | {tot['mem'] / S:.1f} | {tot['wait'] / S:.1f} | {t / S:.1f} | 100 % |")
