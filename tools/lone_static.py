#!/usr/bin/env python3
"""Static instruction counts of the parts of lone_kernel<false> (cvx_lone.h): compiles cvx_lone.hip (at its -O3, Makefile) with -DCVX_LONE_MARK (comment markers in the assembly) and counts
the instructions between consecutive markers, per copy.  python3 tools/lone_static.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cpuvox_amd", "csrc")
out = "/tmp/lone_static.s"
flags = ["-std=c++17", "-Os", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-enable-post-misched=0", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", f"-I{ROOT}/include", f"-I{SRC}", f"-I{SRC}/host", "--cuda-device-only", "-S", "-o", out, "-DCVX_LONE_MARK", "-O3"]
subprocess.check_call(["hipcc"] + flags + sys.argv[1:] + [os.path.join(SRC, "cvx_lone.hip")], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN4cvxk11lone_kernelILb0E") and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
cur, n = "entry", {"valu": 0, "salu": 0, "mem": 0, "nop": 0}
rows = []
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r"; LMARK (\w+)", s)
    if m:
        rows.append((cur, m.group(1), dict(n)))
        cur, n = m.group(1), {"valu": 0, "salu": 0, "mem": 0, "nop": 0}
        continue
    if not s or s[0] in ".;" or s.endswith(":"):
        continue
    op = s.split()[0]
    if op == "s_nop":
        n["nop"] += 1
    elif op.startswith("v_"):
        n["valu"] += 1
    elif op.startswith("s_"):
        n["salu"] += 1
    else:
        n["mem"] += 1
for a, b, c in rows:
    print(f"{a:20s} -> {b:20s} valu {c['valu']:4d} salu {c['salu']:4d} mem {c['mem']:3d} nop {c['nop']:3d}  total {sum(c.values()):4d}")
