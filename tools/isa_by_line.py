"""Diagnostic: static instruction counts of render_kernel<false> per source line of cvx_kernels.h.
Usage: python tools/isa_by_line.py [extra hipcc flags...]   (cross-compiles with -g -S; no GPU needed)"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cpuvox_amd", "csrc")
out = "/tmp/isa_by_line.s"
flags = ["-std=c++17", "-Os", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-enable-post-misched=0", "-g", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
         "-fno-gpu-flush-denormals-to-zero", f"-I{ROOT}/include", f"-I{SRC}", f"-I{SRC}/host", "--cuda-device-only", "-S", "-o", out]
subprocess.check_call(["hipcc"] + flags + sys.argv[1:] + [os.path.join(SRC, "cvx_gpu.hip")], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN4cvxk13render_kernelILb0E") and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
ft = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        ft[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
cur = None
cnt = collections.defaultdict(collections.Counter)
tot = collections.Counter()
for l in lines[start:end]:
    s = l.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
    if m:
        cur = (ft.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    if not s or s.startswith(".") or s.startswith(";") or s.endswith(":"):
        continue
    op = s.split()[0]
    kind = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
    cnt[cur][kind] += 1
    tot[kind] += 1
    if op in ("v_rcp_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_div_fixup_f32"):
        tot[op] += 1
print(dict(tot))
for key in sorted(k for k in cnt if k and "cvx_kernels" in k[0]):
    c = cnt[key]
    print(f"{key[1]:5d}  valu {c['valu']:4d}  salu {c['salu']:4d}  vmem {c['vmem']:3d}  lds {c['lds']:3d}")
