"""CPU: the C-ABI libraries load and export every symbol their headers declare (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from cpuvox_amd import gpu, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_gpu_library_exports_every_declared_symbol():
    L = gpu.lib()
    declared = _declared("cpuvox_gpu.h", "cvx_")
    assert declared, "no declarations parsed"
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(gpu.EXPORTS) == declared, "cpuvox_amd.gpu.EXPORTS out of sync with include/cpuvox_gpu.h"
    assert b"gfx950" in L.cvx_version()


def test_product_abi_has_no_diagnostic_entry_points():
    """The drop-in boundary is include/cpuvox_gpu.h; section profiles, the occupancy query and the arithmetic self-test live in
    include/cpuvox_gpu_diag.h and in the experiment / profiling builds only."""
    header = open(os.path.join(ROOT, "include", "cpuvox_gpu.h")).read()
    assert "debug" not in header.lower() and "selftest" not in header.lower()
    diag = _declared("cpuvox_gpu_diag.h", "cvx_")
    assert sorted(gpu.DIAG_EXPORTS) == diag
    L = gpu.lib()
    assert not [n for n in diag if hasattr(L, n)], "the product library exports diagnostics"
    data = open(os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu.so"), "rb").read()
    assert b"cvx_debug" not in data and b"selftest" not in data
    exp = os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu_exp.so")
    if os.path.exists(exp):
        E = C.CDLL(exp)
        assert not [n for n in diag + _declared("cpuvox_gpu.h", "cvx_") if not hasattr(E, n)], "the experiment build exports the product ABI plus the diagnostics"


def test_host_library_exports_every_declared_symbol():
    L = host.lib()
    declared = _declared("cpuvox_host.h", "cvxh_")
    missing = [n for n in declared if not hasattr(L, n)]
    assert declared and not missing, missing


def test_struct_layouts_match_the_reference_blittable_layout():
    # RenderManager.SegmentData (RenderManager.cs:503-510): 4 float2 + int = 36 bytes
    assert C.sizeof(host.SegmentData) == 36
    # CameraData (CameraData.cs:11-16): float4x4 + float2 + float + bool(+3) + float + float[6] = 108 bytes
    assert C.sizeof(host.CameraData) == 108
    assert host.CameraData.InverseElementIterationDirection.offset == 76
    assert host.CameraData.FarClip.offset == 80 and host.CameraData.LODDistances.offset == 84


def test_gpu_code_object_is_gfx950():
    data = open(gpu.lib_path(), "rb").read()
    assert b"gfx950" in data


def test_product_library_reads_no_environment_and_holds_one_render_kernel():
    """`strings libcpuvox_gpu.so | grep CVX_` is empty: no diagnostic environment switch, no experiment kernel in the shipped library
    (those live in the -DCVX_EXPERIMENTS build, libcpuvox_gpu_exp.so, which the tests of the switches load explicitly)."""
    data = open(os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu.so"), "rb").read()
    assert b"CVX_" not in data
    assert b"getenv" not in data, "the product library must not look at the caller's environment"
    assert b"render_sm_kernel" not in data
    assert data.count(b"_ZN4cvxk13render_kernelILb0E") > 0 and data.count(b"_ZN4cvxk13render_kernelILb1E") > 0
    exp = os.path.join(ROOT, "cpuvox_amd", "libcpuvox_gpu_exp.so")
    if os.path.exists(exp):
        assert b"CVX_TILE_SPLIT" in open(exp, "rb").read()


def test_product_fails_loudly_without_a_device():
    """No CPU fallback: without a HIP device context creation raises (on a GPU box it succeeds)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(gpu.CvxError):
        gpu.Context(0)


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "cpuvox_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", "Makefile")):
                text = open(os.path.join(root, f), errors="ignore").read()
                assert "oraclelib" not in text and "cvx_oracle" not in text and "libcvx_oracle" not in text, os.path.join(root, f)


@pytest.mark.parametrize("world_size", [1, 2, 3, 8])
def test_native_shard_plan_matches_the_python_plan(world_size):
    """cvx_shard_plan_* (the multi-GPU plan behind the C ABI, no GPU needed) against cpuvox_amd.dist.ShardPlan: same tile
    count, same section boundaries, same output address for every tile, for every rank."""
    import ctypes as C

    import numpy as np

    import scenes
    from cpuvox_amd import dist as cdist

    W, H = 640, 480
    ws = scenes.load_world("proc256")
    frames = [scenes.benchmark_frame(ws, W, H, t, 4.0) for t in (0.0, 0.2, 0.45, 0.6, 0.75, 0.8, 0.9, 1.0, 1.1, 0.33, 0.66)]
    n = len(frames)
    segs = (host.SegmentData * (4 * n))()
    cams = (host.CameraData * n)()
    vps = (C.c_float * (2 * n))()
    for i, f in enumerate(frames):
        for s in range(4):
            segs[4 * i + s] = f.segments[s]
        cams[i] = f.camera
        vps[2 * i], vps[2 * i + 1] = f.vanishingPointScreenSpace[0], f.vanishingPointScreenSpace[1]
    packed = (n, segs, cams, vps)
    send_base, disp_base = 0x7F0000000000, 0x7E0000000000
    rendered = np.zeros(0, dtype=np.int64)
    for rank in range(world_size):
        ref = cdist.ShardPlan(frames, W, H, rank, world_size)
        nat = gpu.NativeShardPlan(packed, W, H, rank, world_size)
        assert nat.tile_count == ref.tile_count
        assert np.array_equal(nat.send_start, ref.send_start) and np.array_equal(nat.disp_start, ref.disp_start)
        a, b = nat.tile_out(send_base, disp_base), ref.tile_out(send_base, disp_base)
        assert np.array_equal(a, b)
        rendered = np.concatenate([rendered, np.flatnonzero(a)])
        # what cvx_exchange sends / receives per peer == the slices the Python exchange hands to isend / irecv, and the two sides of
        # every pair agree on the size (rank r sends to p exactly what p expects from r)
        for peer in range(world_size):
            s0, sn, r0, rn = nat.transfer(peer)
            if peer == rank:
                assert (sn, rn) == (0, 0)
                continue
            assert (s0, sn) == (int(ref.send_start[peer]), int(ref.send_start[peer + 1] - ref.send_start[peer]))
            assert (r0, rn) == (int(ref.disp_start[peer]), int(ref.disp_start[peer + 1] - ref.disp_start[peer]))
            other = cdist.ShardPlan(frames, W, H, peer, world_size)
            assert sn == int(other.disp_start[rank + 1] - other.disp_start[rank]) and rn == int(other.send_start[rank + 1] - other.send_start[rank])
        nat.close()
    # every tile of the batch is rendered by exactly one rank
    assert np.array_equal(np.sort(rendered), np.arange(ref.tile_count))


def test_exchange_entry_points_fail_cleanly_without_a_device_or_peers():
    """cvx_exchange / cvx_comm_*: argument errors are reported, nothing is dereferenced (the data path needs >= 2 GPUs)."""
    L = gpu.lib()
    assert L.cvx_exchange(None, None, None, None, None, None) == -1      # CVX_ERR_INVALID_ARGUMENT: no context
    assert L.cvx_comm_create(None, None, 0, 1, None) == -1
    assert L.cvx_comm_unique_id(None) == -1
    assert L.cvx_comm_destroy(None) == 0
    assert L.cvx_shard_plan_tile_count(None) == 0


def test_bench_attaches_the_committed_counter_summary_of_this_library():
    """bench.py's `roofline.traffic` comes from profiles/rNN_pmc_render_kernel*.csv, matched by the sha-256 of the library it loads and by the workload
    arguments.  When the tree's library is the one the newest profiles were collected with, the default workload must find its file
    (round 4: the lookup referenced a name that did not exist at module level and silently returned None)."""
    import argparse
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sys_path = os.path.join(ROOT, "tools")
    import sys
    sys.path.insert(0, sys_path)
    from pmc_aggregate import library_sha256

    newest = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_pmc_render_kernel.csv"))[-1]
    stamp = json.loads(open(os.path.join(ROOT, "profiles", newest)).readline()[1:])
    args = argparse.Namespace(frames=512, width=1920, height=1080, world="proc2048", lod_error=1.0, pose_range=None, steps=10, warmup=2)
    found = bench.find_counter_summary(args)
    if stamp.get("library_sha256") == library_sha256(gpu.lib_path()):
        assert found and os.path.basename(found) == newest, found
    else:
        assert found is None or os.path.basename(found) != newest  # another build: its counters are not this library's


def test_counter_summary_serves_the_drivers_argument_vector(tmp_path, monkeypatch):
    """the driver runs `bench.py --gpus 1 --steps 20 --warmup 5`, the builder's default is 10 / 2 -- and round 4's summary,
    collected with the default, was refused for the driver's line (`roofline.traffic: null`).  A summary now lists every launch, bench.py takes
    the mean over the timed steps of ITS run; both argument vectors must find the file, a longer run than the collected one must not."""
    import importlib.util
    import json
    import sys

    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from pmc_aggregate import library_sha256

    (tmp_path / "profiles").mkdir()
    stamp = {"kernel_sources_sha256": "x", "library_sha256": library_sha256(gpu.lib_path()),
             "bench_args": ["--cpu-seconds", "0", "--latency-frames", "0", "--steps", "20", "--warmup", "5"]}
    fetch = [1000.0 + i for i in range(25)]
    write = [10.0 * (i % 2) for i in range(25)]
    with open(tmp_path / "profiles" / "r99_pmc_render_kernel.csv", "w") as fh:
        fh.write("# " + json.dumps(stamp) + "\n")
        fh.write("counter,mean_per_launch,launches,per_launch\n")
        fh.write("FETCH_SIZE,%g,25,%s\n" % (sum(fetch) / 25, ";".join("%.9g" % v for v in fetch)))
        fh.write("WRITE_SIZE,%g,25,%s\n" % (sum(write) / 25, ";".join("%.9g" % v for v in write)))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.delenv("CVX_GPU_LIB", raising=False)

    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "1", "--steps", "20", "--warmup", "5"])  # the driver's command line, verbatim
    args = bench.parse_args()
    found = bench.find_counter_summary(args)
    assert found and found.endswith("r99_pmc_render_kernel.csv"), found
    c = bench.read_counter_summary(found, args.warmup, args.steps)
    assert c["FETCH_SIZE"] == sum(fetch[5:25]) / 20 and c["WRITE_SIZE"] == sum(write[5:25]) / 20

    monkeypatch.setattr(sys, "argv", ["bench.py"])  # the default run: steps 2 .. 11 of the same launches
    args = bench.parse_args()
    found = bench.find_counter_summary(args)
    assert found, "the default --steps 10 --warmup 2 is covered by the 25 collected launches"
    assert bench.read_counter_summary(found, args.warmup, args.steps)["FETCH_SIZE"] == sum(fetch[2:12]) / 10

    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "30", "--warmup", "5"])  # longer than what was collected: no counters for steps 25 ..
    assert bench.find_counter_summary(bench.parse_args()) is None
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "20", "--warmup", "5", "--width", "3840", "--height", "2160"])  # another workload
    assert bench.find_counter_summary(bench.parse_args()) is None
    # a summary of rounds 1-4 (no per-launch column) still serves exactly the run that collected it, nothing else
    old = os.path.join(ROOT, "profiles", "r04_pmc_render_kernel.csv")
    assert bench.read_counter_summary(old, 2, 10) is not None and bench.read_counter_summary(old, 5, 20) is None


def test_gpu_objects_do_not_depend_on_the_build_directory(tmp_path):
    """bench.py attaches the committed counter summary only to the library whose sha-256 it is stamped with, and the GPU box rebuilds stale targets at
    its own scratch path: the product library's objects are compiled with a fixed compilation-unit id (csrc/Makefile), so the same sources give the
    same bytes in any directory.  Checked on the smallest translation unit, built through the Makefile's own rule in two places."""
    import hashlib
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("hipcc") is None:
        pytest.skip("no hipcc here")
    digests = []
    for name in ("a", "somewhere/else"):
        tree = tmp_path / name
        shutil.copytree(os.path.join(root, "cpuvox_amd", "csrc"), tree / "cpuvox_amd" / "csrc", ignore=shutil.ignore_patterns(".obj", "*.o", "*.so"))
        shutil.copytree(os.path.join(root, "include"), tree / "include")
        obj = tree / "cpuvox_amd" / "csrc" / ".obj" / "product" / "cvx_shard.o"
        subprocess.run(["make", "-C", str(tree / "cpuvox_amd" / "csrc"), str(obj)], check=True, capture_output=True, text=True, timeout=600)
        digests.append(hashlib.sha256(obj.read_bytes()).hexdigest())
    assert digests[0] == digests[1]
