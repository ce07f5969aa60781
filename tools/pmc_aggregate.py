#!/usr/bin/env python3
"""Aggregate the per-pass rocprofv3 --pmc outputs of tools/pmc_passes.sh into one small CSV.

usage: tools/pmc_aggregate.py <pmc outdir> <kernel substring> [bench args of the passes...] > profiles/rNN_pmc_render_kernel.csv
(first line: a comment with the sha-256 of the kernel sources and the bench arguments the counters were collected with)

For every counter: mean over the dispatches of kernels whose name contains the substring (counter values of one
dispatch are summed over the rows rocprofv3 emits for it, e.g. one row per XCD/instance), and -- fourth column -- the value of
every single launch in dispatch order (`;`-separated).  Launch i of a bench.py run is its step i (warm-up steps first), and
which frames a step renders depends on its index only, so bench.py can assemble the mean over the timed steps of ANY
--steps / --warmup that the collected launches cover (round 5: the driver runs --steps 20 --warmup 5, the default is 10 / 2)."""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["cvx_kernels.h", "cvx_lone.h", "cvx_lone.hip", "cvx_device.h", "cvx_context.h", "cvx_downsample.h", "cvx_gpu.hip", "cvx_world.hip", "cvx_shard.hip", "Makefile"]


def kernel_sources_sha256() -> str:
    """Identity of the build a counter file belongs to: the sources libcpuvox_gpu.so is compiled from (bench.py only attaches
    counter traffic to its line when this matches the tree it runs from)."""
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "cpuvox_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()


def library_sha256(path=None) -> str:
    """Identity of the library the counters were collected with: the file itself (the build is deterministic: same sources and flags, same bytes)."""
    path = path or os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu.so")
    with open(path, "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()


def main():
    outdir, needle = sys.argv[1], sys.argv[2]
    per_dispatch = defaultdict(float)
    for path in sorted(glob.glob(f"{outdir}/pass*/**/*_counter_collection.csv", recursive=True)):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if needle not in row["Kernel_Name"]:
                    continue
                per_dispatch[(path, row["Dispatch_Id"], row["Counter_Name"])] += float(row["Counter_Value"])
    sums, counts, series = defaultdict(float), defaultdict(int), defaultdict(list)
    for (path, dispatch, name), v in per_dispatch.items():
        sums[name] += v
        counts[name] += 1
        series[name].append((path, int(dispatch), v))
    stamp = {"kernel_sources_sha256": kernel_sources_sha256(), "library_sha256": library_sha256(), "bench_args": sys.argv[3:]}
    print("# " + json.dumps(stamp))
    print("counter,mean_per_launch,launches,per_launch")
    for name in sums:
        per_launch = ";".join(f"{v:.9g}" for _, _, v in sorted(series[name]))  # (one pass = one process: dispatch ids ascend in launch order)
        print(f"{name},{sums[name] / counts[name]:.6g},{counts[name]},{per_launch}")


if __name__ == "__main__":
    main()
