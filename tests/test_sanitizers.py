"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer builds of the two CPU-side native pieces, exercised in child processes
(GPU sanitizers are not available on this pool; the HIP side is covered by upload-time validation and the parity tests).

  * oracle/libcvx_oracle_asan.so: three scenes (single segment from outside the world, four segments, deep LODs) rendered
    through the sanitized oracle; the raybuffers must equal the regular build's.
  * cpuvox_amd/libcpuvox_host_asan.so: world building (procedural, mill.obj voxelisation fixture round trip, LOD chain),
    .world save / load, camera / segment setup and the rejection of malformed files."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _asan_runtime():
    path = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def _run(code, extra_env):
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("libasan runtime not found")
    env = dict(os.environ)
    env.update(extra_env)
    env["LD_PRELOAD"] = rt  # the interpreter is not instrumented: the runtime has to come first
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["OMP_NUM_THREADS"] = "4"
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "tests")])
    r = subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    assert r.returncode == 0, f"sanitized run failed ({r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
    return r.stdout


def test_oracle_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libcvx_oracle_asan.so"], stdout=subprocess.DEVNULL)
    import scenes

    names = ["mill256_t0", "mill256_t075", "proc256_low_lod10"]
    code = f"""
        import json, scenes, oraclelib as O
        out = {{}}
        for name in {names!r}:
            ws, fr, W, H = scenes.scene_frame(name)
            td, lr, cnt = O.draw_segments(ws, fr, W, H, threads=4)
            out[name] = [scenes.crc(td), scenes.crc(lr), cnt.S, cnt.P]
        print("RESULT", json.dumps(out))
    """
    stdout = _run(code, {"CVX_ORACLE_LIB": os.path.join(ROOT, "oracle", "libcvx_oracle_asan.so")})
    import json

    import oraclelib as O

    got = json.loads(stdout.split("RESULT", 1)[1])
    for name in names:
        ws, fr, W, H = scenes.scene_frame(name)
        td, lr, cnt = O.draw_segments(ws, fr, W, H)
        assert got[name] == [scenes.crc(td), scenes.crc(lr), cnt.S, cnt.P], name


def test_host_library_under_asan_and_ubsan(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "cpuvox_amd", "csrc"), "host-asan"], stdout=subprocess.DEVNULL)
    code = f"""
        import os, numpy as np, scenes
        from cpuvox_amd import host
        ws = host.WorldSet.procedural(128, 64, 256, 7)
        assert ws.lod_count == 6
        path = {str(tmp_path / "w.world")!r}
        ws.save(path)
        back = host.WorldSet.load(path)
        for lod in range(6):
            assert np.array_equal(back.storage(lod), ws.storage(lod))
        mill = scenes.load_world("mill256")          # .world fixture (mill.obj voxelised): load path + validation
        fr = scenes.benchmark_frame(mill, 640, 480, 0.75)
        assert fr.totalRays > 0
        rebuilt = host.WorldSet.from_blobs(ws.dims, [ws.storage(l) for l in range(6)])
        assert rebuilt.lod_count == 6
        # malformed inputs are refused, not walked
        raw = bytearray(open(path, "rb").read())
        for cut in (10, 40, len(raw) // 2):
            open(path + ".bad", "wb").write(raw[:cut])
            try:
                host.WorldSet.load(path + ".bad")
                raise SystemExit("truncated file accepted")
            except RuntimeError:
                pass
        bad = bytearray(raw)
        table = 24 + 16 * 6                            # header + (offset, length) table
        bad[table + 0: table + 4] = (0x7FFFFFF0).to_bytes(4, "little")   # first column header: element offset far outside the pool
        bad[table + 4: table + 6] = (3).to_bytes(2, "little")
        open(path + ".bad", "wb").write(bad)
        try:
            host.WorldSet.load(path + ".bad")
            raise SystemExit("file with an out-of-range column accepted")
        except RuntimeError:
            pass
        # the JPEG decoder (map_Kd textures) on untrusted bytes: truncated, bit-flipped and progressive streams, an Adobe RGB marker
        try:
            from PIL import Image
        except ImportError:
            Image = None
        if Image is not None:
            rng = np.random.default_rng(11)
            img = Image.fromarray(rng.integers(0, 255, (37, 53, 3), dtype=np.uint8))
            jpg = {str(tmp_path / "t.jpg")!r}
            for opts in (dict(quality=90, subsampling=2), dict(quality=85, subsampling=0, progressive=True)):
                img.save(jpg, "JPEG", **opts)
                good = bytearray(open(jpg, "rb").read())
                assert host.load_image(jpg).shape == (37, 53, 4)
                cases = [good[:cut] for cut in (2, 4, 20, len(good) // 3, len(good) // 2, len(good) - 2)]
                for k in range(60):
                    b = bytearray(good)
                    for _ in range(1 + k % 4):
                        b[int(rng.integers(2, len(b)))] ^= 1 << int(rng.integers(0, 8))
                    cases.append(b)
                sos = good.find(b"\\xff\\xda")
                cases.append(good[:sos + 4])                      # the file ends inside the SOS header
                for b in cases:
                    open(jpg, "wb").write(b)
                    try:
                        host.load_image(jpg)                          # either an image or a clean refusal; never a sanitizer report
                    except RuntimeError:
                        pass
            img.save(jpg, "JPEG", quality=90)
            good = bytearray(open(jpg, "rb").read())
            app14 = b"\\xff\\xee\\x00\\x0eAdobe\\x00\\x64\\x00\\x00\\x00\\x00\\x00"    # transform 0 = RGB: refused, not decoded with wrong colours
            open(jpg, "wb").write(good[:2] + app14 + good[2:])
            try:
                host.load_image(jpg)
                raise SystemExit("Adobe RGB JPEG accepted")
            except RuntimeError as e:
                assert "colour transform" in str(e), e
        print("RESULT ok")
    """
    stdout = _run(code, {"CVX_HOST_LIB": os.path.join(ROOT, "cpuvox_amd", "libcpuvox_host_asan.so")})
    assert "RESULT ok" in stdout
