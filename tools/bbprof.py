#!/usr/bin/env python3
"""Basic-block execution profile of render_kernel<false> without a thread-trace decoder (none is installed on this pool).

The device assembly of libcpuvox_gpu is compiled once; for every basic block of the kernel a variant is assembled with one
`s_nop 0` inserted at the top of that block (nothing else changes: same registers, same arithmetic, same pictures).  A driver
process loads the variants one after the other and renders the same frames once with each; `rocprofv3 --pmc SQ_INSTS_SALU`
around that process reports the scalar-instruction count of every dispatch, and (count of variant b) - (count of the
unmodified build) is exactly how often block b was executed by a wave.  Executions x the block's static instruction counts
= the dynamic instruction budget per block; the column sums reproduce SQ_INSTS_VALU / SQ_INSTS_SALU of the plain kernel,
which is the self-check printed at the end.

  python tools/bbprof.py build <workdir> [-j N]      compile the base + all variants (CPU only; hipcc needed)
  python tools/bbprof.py drive <workdir> [frames]    (run under rocprofv3 --pmc ...) one dispatch per variant
  python tools/bbprof.py report <workdir> <pmc csv dir>   table: block, source lines, static counts, executions, dynamic counts
  tools/bbprof.sh <tag>                              all three on the GPU box, results under gpurun_out/<tag>/
"""
from __future__ import annotations

import collections
import concurrent.futures
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cpuvox_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
KERNEL = "_ZN4cvxk13render_kernelILb0EEEvPK8DevFramePK7DevTilePK8DevWorldP11DevCounters"
HIPFLAGS = ["-std=c++17", "-Os", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp", "-mllvm", "-enable-post-misched=0", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
            "-fno-gpu-flush-denormals-to-zero", f"-I{ROOT}/include", f"-I{SRC}/host", f"-I{SRC}"]


def run(cmd, **kw):
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, **kw)


def classify(op: str) -> str:
    if op.startswith("s_cbranch") or op in ("s_branch", "s_setpc_b64", "s_endpgm"):
        return "branch"
    if op.startswith("s_waitcnt") or op == "s_nop":
        return "wait"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_dcache"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    return "vmem"


HALF_RATE = re.compile(r"^v_(cmp|cmpx|cndmask|min|max|med3|cvt|floor|ceil|rndne|trunc|fract|mul_lo|mul_hi|mad_u|mad_i|lshl_add|add_lshl|lshl_or|and_or|or3|xad|bfe|bfi|ffb|div_|rcp|rsq|sqrt|bitop|perm|readlane|readfirstlane|mbcnt|pk_)")


def parse_blocks(lines):
    """Blocks of the kernel function: (first line index of the block header, label, [instruction lines], {source lines})."""
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks = []
    cur = {"line": start, "label": "entry", "ins": [], "src": collections.Counter(), "depth": ""}
    loc = None
    for i in range(start + 1, end):
        l = lines[i]
        s = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %bb\.(\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"line": i, "label": m.group(1), "ins": [], "src": collections.Counter(), "depth": ""}
            continue
        m = re.match(r"\.loc\s+\d+\s+(\d+)", s)
        if m:
            loc = int(m.group(1))
            continue
        if not s or s.startswith(".") or s.startswith(";") or s.endswith(":"):
            continue
        op = s.split()[0]
        cur["ins"].append(op)
        if loc:
            cur["src"][loc] += 1
    blocks.append(cur)
    return blocks


def cmd_build(work, jobs):
    os.makedirs(work, exist_ok=True)
    base = os.path.join(work, "base")
    os.makedirs(base, exist_ok=True)
    # 1. base objects + device assembly (line tables only: same code as the product build, plus .loc for the report)
    run(["hipcc"] + HIPFLAGS + ["-gline-tables-only", "-c", "-save-temps", os.path.join(SRC, "cvx_gpu.hip"), "-o", "cvx_gpu.o"], cwd=base)
    for f in ("cvx_world", "cvx_shard"):
        run(["hipcc"] + HIPFLAGS + ["-c", os.path.join(SRC, f + ".hip"), "-o", f + ".o"], cwd=base)
    dev_s = os.path.join(base, "cvx_gpu-hip-amdgcn-amd-amdhsa-gfx950.s")
    lines = open(dev_s).read().split("\n")
    blocks = parse_blocks(lines)
    meta = []
    for b, blk in enumerate(blocks):
        c = collections.Counter(classify(op) for op in blk["ins"])
        half = sum(1 for op in blk["ins"] if HALF_RATE.match(op))
        src = sorted(blk["src"].items(), key=lambda kv: -kv[1])
        meta.append({"block": b, "label": blk["label"], "line": blk["line"], "static": dict(c), "half_rate_valu": half, "n": len(blk["ins"]),
                     "src": [k for k, _ in src[:6]], "src_min": min(blk["src"]) if blk["src"] else 0, "src_max": max(blk["src"]) if blk["src"] else 0})
    json.dump(meta, open(os.path.join(work, "blocks.json"), "w"))

    def build_variant(b):
        name = "base" if b < 0 else f"b{b:04d}"
        d = os.path.join(work, "v", name)
        os.makedirs(d, exist_ok=True)
        out = list(lines)
        if b >= 0:
            at = blocks[b]["line"] + 1
            out.insert(at, "\ts_mov_b32 vcc_lo, vcc_lo")  # counted by SQ_INSTS_SALU (s_nop is not), changes nothing
        s_path = os.path.join(d, "dev.s")
        open(s_path, "w").write("\n".join(out))
        run([f"{LLVM}/clang", "-cc1as", "-triple", "amdgcn-amd-amdhsa", "-filetype", "obj", "-target-cpu", "gfx950", "-mrelocation-model", "pic", "-o", "dev.o", "dev.s"], cwd=d)
        run([f"{LLVM}/lld", "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", "dev.out", "dev.o"], cwd=d)
        run([f"{LLVM}/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
             "-input=/dev/null", "-input=dev.out", "-output=dev.hipfb"], cwd=d)
        run(["objcopy", "--update-section", ".hip_fatbin=dev.hipfb", os.path.join(base, "cvx_gpu.o"), "host.o"], cwd=d)
        so = os.path.join(work, "libs", f"lib_{name}.so")
        run(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so, "host.o", os.path.join(base, "cvx_world.o"), os.path.join(base, "cvx_shard.o"), "-ldl"], cwd=d)
        for f in ("dev.s", "dev.o", "dev.out", "dev.hipfb", "host.o"):
            os.remove(os.path.join(d, f))
        return name

    os.makedirs(os.path.join(work, "libs"), exist_ok=True)
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        done = list(ex.map(build_variant, range(-1, len(blocks))))
    print(f"{len(done)} libraries ({len(blocks)} blocks) under {work}/libs")


def cmd_drive(work, frames):
    """One dispatch of render_kernel<false> per library, in the order base, b0000, b0001, ... (run under rocprofv3 --pmc)."""
    os.environ.setdefault("CVX_NO_TORCH_PRELOAD", "1")
    sys.path.insert(0, ROOT)
    from cpuvox_amd import gpu, host

    W, H = 1920, 1080
    ws = host.WorldSet.procedural(2048, 2048, 2048, 0x5EED2048)
    pose0 = host.camera_pose((0, 0, 0), (0, 0, 0), W, H)
    lods, far = host.setup_lods(pose0, ws.max_dimension, W, H, 1.0)
    fr = []
    for g in range(frames):
        i = (g * 37) % 1000
        pos, eul = host.sample_benchmark_path(i / 1000 * host.BENCHMARK_PATH_LENGTH, ws.dims)
        fr.append(host.setup_frame(host.camera_pose(pos, eul, W, H), lods, far, W, H, ws.dims[1]))
    libs = sorted(glob.glob(os.path.join(work, "libs", "lib_*.so")), key=lambda p: (0 if p.endswith("lib_base.so") else 1, p))
    order = []
    for path in libs:
        gpu.use_library(path)
        ctx = gpu.Context(0, buffer_count=frames)
        ctx.upload_world(ws)
        ctx.set_resolution(W, H)
        ctx.enable_counters(False)
        ctx.draw_packed(ctx.pack_batch(fr), 0, gpu.DRAW_SYNC)
        ctx.close()
        order.append(os.path.basename(path))
        if len(order) % 25 == 0:
            print(f"{len(order)} / {len(libs)}", flush=True)
    json.dump(order, open(os.path.join(work, "order.json"), "w"))


def cmd_report(work, pmc_dir):
    meta = json.load(open(os.path.join(work, "blocks.json")))
    order = json.load(open(os.path.join(work, "order.json")))
    rows = collections.defaultdict(dict)  # dispatch id -> counter -> value
    for p in glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(p)):
            if "render_kernel<false>" in r["Kernel_Name"]:
                rows[int(r["Dispatch_Id"])][r["Counter_Name"]] = rows[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(rows)
    # a draw is one launch per LDS class (round 4): the same number of consecutive dispatches for every library, summed
    assert len(ids) % len(order) == 0, (len(ids), len(order))
    per = len(ids) // len(order)
    draws = []
    for k in range(len(order)):
        tot_ = collections.Counter()
        for i in ids[k * per:(k + 1) * per]:
            tot_.update(rows[i])
        draws.append(dict(tot_))
    base = draws[0]
    execs = {}
    for k, name in enumerate(order[1:], start=1):
        execs[int(name[5:9])] = draws[k]["SQ_INSTS_SALU"] - base["SQ_INSTS_SALU"]
    steps = max(execs.values())  # the loop header block of the busier instance... printed for orientation only
    tot = collections.Counter()
    out = []
    for m in meta:
        e = execs.get(m["block"], 0.0)
        st = m["static"]
        dyn = {k: e * st.get(k, 0) for k in ("valu", "salu", "branch", "vmem", "lds", "smem", "wait")}
        for k, v in dyn.items():
            tot[k] += v
        out.append((e * (m["n"] - st.get("wait", 0)), m, e, dyn))
    out.sort(key=lambda t: -t[0])
    total_dyn = sum(t[0] for t in out)
    print(f"# basic-block profile of render_kernel<false>: {len(meta)} blocks, base dispatch SALU {base['SQ_INSTS_SALU']:.4g}")
    print(f"# self-check: sum over blocks  VALU {tot['valu']:.5g}  SALU {tot['salu'] + tot['branch'] + tot['wait']:.5g} (+ s_waitcnt / s_nop counted as SALU by the hardware: {tot['wait']:.4g})")
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU"):
        if k in base:
            print(f"#             measured         {k} {base[k]:.5g}")
    print("# block label        asm line  src lines (most frequent first)        static v/s/br/mem  executions   dyn instr   share   cumulative")
    cum = 0.0
    for d, m, e, dyn in out[:120]:
        cum += d
        st = m["static"]
        print(f"{m['block']:4d} {m['label']:14s} {m['line']:6d}  {str(m['src'][:5]):38s} {st.get('valu', 0):3d}/{st.get('salu', 0):3d}/{st.get('branch', 0):2d}/{st.get('vmem', 0) + st.get('lds', 0):2d} "
              f"{e:12.0f} {d:12.4g} {100 * d / total_dyn:6.2f}% {100 * cum / total_dyn:6.1f}%")
    json.dump({"executions": execs, "base": base}, open(os.path.join(work, "profile.json"), "w"))


if __name__ == "__main__":
    cmd = sys.argv[1]
    if cmd == "build":
        jobs = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else 8
        cmd_build(sys.argv[2], jobs)
    elif cmd == "drive":
        cmd_drive(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 32)
    elif cmd == "report":
        cmd_report(sys.argv[2], sys.argv[3])
