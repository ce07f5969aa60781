#!/usr/bin/env python3
"""Device-assembly pass of the libcpuvox_gpu build (gfx950 only).

hipcc compiles cvx_gpu.hip to device assembly (-save-temps), this script re-encodes a few instructions whose compiler-chosen form is slow on
gfx950 (measured with tools/valu_rate.hip, profiles/r04_valu_rate.txt), assembles and links the code object again and puts it back into the
host object as its .hip_fatbin.  Every pass is a pure re-encoding: same operation on the same operands, so results cannot change (and the
parity suite runs on the library this produces).

  python3 asm_pass.py --out <lib.so> --passes cnd64[,...] [--work DIR] -- hipcc <flags...>

Passes:
  cnd64   v_cndmask_b32_e32 d, a, b, vcc  ->  v_cndmask_b32_e64 d, a, b, vcc
          The VOP2 select (mask implied in VCC) issues at ~16 cycles per instruction whenever selects follow one another (and 19 when VCC came
          from the scalar unit); the VOP3 form with the same VCC as an explicit operand issues at the normal 4.  +4 bytes of code per select.
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
LLVM = "/opt/rocm/lib/llvm/bin"
OTHER_SOURCES = ["cvx_lone.hip", "cvx_world.hip", "cvx_shard.hip"]


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout[-4000:] + "\n")
        raise SystemExit(r.returncode)
    return r.stdout


def pass_cnd64(lines, stats):
    pat = re.compile(r"^(\s*)v_cndmask_b32_e32(\s+)(v\d+),\s*([^,]+),\s*(v\d+),\s*vcc\s*$")
    for i, l in enumerate(lines):
        m = pat.match(l)
        if m:
            lines[i] = f"{m.group(1)}v_cndmask_b32_e64{m.group(2)}{m.group(3)}, {m.group(4)}, {m.group(5)}, vcc"
            stats["cnd64"] = stats.get("cnd64", 0) + 1


PASSES = {"cnd64": pass_cnd64}


def transform(text, passes, stats):
    lines = text.split("\n")
    for p in passes:
        PASSES[p](lines, stats)
    return "\n".join(lines)


def build(out_so, work, passes, compiler):
    """compiler = ["hipcc", flags...] exactly as the plain build uses them."""
    os.makedirs(work, exist_ok=True)
    run(compiler + ["-c", "-save-temps", os.path.join(HERE, "cvx_gpu.hip"), "-o", "cvx_gpu.o"], work)
    objs = []
    for src in OTHER_SOURCES:
        o = os.path.splitext(src)[0] + ".o"
        run(compiler + ["-c", os.path.join(HERE, src), "-o", o], work)
        objs.append(o)
    dev_s = os.path.join(work, "cvx_gpu-hip-amdgcn-amd-amdhsa-gfx950.s")
    stats = {}
    text = transform(open(dev_s).read(), passes, stats)
    open(os.path.join(work, "dev.s"), "w").write(text)
    run([f"{LLVM}/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-target-cpu", "gfx950", "-mrelocation-model", "pic", "-o", "dev.o", "dev.s"], work)
    run([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", "dev.out", "dev.o"], work)
    run([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
         "-input=/dev/null", "-input=dev.out", "-output=dev.hipfb"], work)
    run(["objcopy", "--update-section", ".hip_fatbin=dev.hipfb", "cvx_gpu.o", "host.o"], work)
    run([compiler[0], "-shared", "-fPIC", "--offload-arch=gfx950", "-o", out_so, "host.o"] + objs + ["-ldl"], work)
    return stats


def main():
    argv = sys.argv[1:]
    if "--" not in argv:
        raise SystemExit(__doc__)
    split = argv.index("--")
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--passes", default="cnd64")
    ap.add_argument("--work", default=None)
    a = ap.parse_args(argv[:split])
    compiler = argv[split + 1:]
    # (-shared / -o of the plain link line do not belong to the compile steps)
    work = a.work or tempfile.mkdtemp(prefix="cvx_asm_")
    stats = build(os.path.abspath(a.out), work, [p for p in a.passes.split(",") if p and p != "none"], compiler)
    print(f"asm_pass: {os.path.basename(a.out)} {stats}")


if __name__ == "__main__":
    main()
