"""The host library's loaders and builders under the host sanitizers (SURVEY §5 "race detection / sanitizers": the reference is
safe single-threaded Rust; this C++ host side has a threaded PLY parser and a threaded SAH builder).  tests/native/host_sanitize.cpp
links scene.cpp + accel.cpp directly (no HIP), loads every scene, builds every BVH mode and grid size, and walks the error paths;
once under ASan + UBSan, once under TSan with the thread thresholds lowered so that the helper threads run on the small meshes too.
GPU sanitizers are not available on this pool; device code is held to the oracle instead."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = [os.path.join(ROOT, "tests/native/host_sanitize.cpp"), os.path.join(ROOT, "raytracer_amd/csrc/host/scene.cpp"),
       os.path.join(ROOT, "raytracer_amd/csrc/host/accel.cpp")]


@pytest.mark.parametrize("name,flags,env", [
    ("asan_ubsan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"], {}),
    ("tsan", ["-fsanitize=thread"], {"RT_PLY_PAR_MIN": "1000", "RT_PLY_THREADS": "4", "RT_BVH_PAR_MIN": "2000"}),
])
def test_host_loaders_and_builders_are_clean_under_sanitizers(tmp_path, name, flags, env):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = str(tmp_path / f"host_{name}")
    cc = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-pthread"] + flags + ["-o", exe] + SRC, capture_output=True, text=True)
    if cc.returncode != 0 and ("cannot find" in cc.stderr or "unrecognized" in cc.stderr):
        pytest.skip(f"this toolchain has no {name} runtime: {cc.stderr[-200:]}")
    assert cc.returncode == 0, cc.stderr[-3000:]
    e = dict(os.environ); e.update(env)
    r = subprocess.run([exe, ROOT], capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ok (0 failures)" in r.stdout and "Sanitizer" not in r.stderr, r.stderr[-3000:]
