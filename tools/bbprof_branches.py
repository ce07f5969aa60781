#!/usr/bin/env python3
"""Branch instructions of the busier trace_ray instance in a basic-block profile, with executions per wave-step and how often the fall-through is taken.
usage: python tools/bbprof_branches.py <dir with blocks.json, profile.json, device.s> [N]"""
import json
import re
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 80
meta = json.load(open(d + "/blocks.json"))
lines = open(d + "/device.s").read().split("\n")
ex = {int(k): v for k, v in json.load(open(d + "/profile.json"))["executions"].items()}


def body(i):
    a = meta[i]["line"]
    z = meta[i + 1]["line"] if i + 1 < len(meta) else len(lines)
    loc = None
    out = []
    for l in lines[a:z]:
        s = l.strip()
        m = re.match(r"\.loc\s+\d+\s+(\d+)", s)
        if m:
            loc = int(m.group(1))
            continue
        if not s or s[0] in ".;" or s.endswith(":"):
            continue
        out.append((loc, s))
    return out


# the column-step heads: blocks with two dwordx4 loads; the busier instance = the pair with most executions
heads = [(ex.get(b["block"], 0), i) for i, b in enumerate(meta) if sum(1 for _, s in body(i) if "global_load_dwordx4" in s) >= 2 and ex.get(b["block"], 0) > 1000]
heads.sort(reverse=True)
steps = heads[0][0] + heads[1][0]
lo = min(heads[0][1], heads[1][1]) - 6
rows = []
for i in range(max(lo, 0), len(meta) - 1):
    bd = body(i)
    if not bd:
        continue
    loc, last = bd[-1]
    if last.startswith("s_cbranch") or last.startswith("s_branch"):
        e = ex.get(meta[i]["block"], 0)
        e1 = ex.get(meta[i + 1]["block"], 0)
        rows.append((e / steps, meta[i]["block"], loc, last, e1 / e if e else 0.0, len(bd)))
rows.sort(reverse=True)
print(f"wave-steps {steps:.0f}; branches per step in the listed range: {sum(r[0] for r in rows):.1f}")
for r in rows[:top]:
    print(f"{r[0]:5.2f}/step  block {r[1]:4d}  line {str(r[2]):>5}  {r[3]:36s} fall-through executed {r[4]:.2f} of the time")
