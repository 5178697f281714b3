"""CPU tests of the host side (no GPU, no compute calls into librt_hip): builders against the oracle's literal
restatement, the CLI surface, and that the C-ABI library loads and exports every declared symbol."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from raytracer_amd import _ffi as F
from raytracer_amd import Camera, HittableList, RtError

from .util import load_world, oracle_of, weekend_mixed


def test_rt_hip_exports_every_declared_symbol():
    """include/rt_hip.h declares it <=> librt_hip.so exports it."""
    hdr = open("include/rt_hip.h").read()
    declared = sorted(set(re.findall(r"\b(rt_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(F.RT_HIP_SYMBOLS)
    lib = C.CDLL(os.path.join(F.LIB_DIR, "librt_hip.so"))
    for s in declared:
        assert getattr(lib, s) is not None
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(F.LIB_DIR, "librt_hip.so")], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (rt_[a-z_]+)$", out, re.M)))
    assert exported == declared


def test_struct_layouts_match_the_header():
    assert C.sizeof(F.RtPrim) == 80 and C.sizeof(F.RtMaterial) == 40 and C.sizeof(F.RtBvhNode) == 72
    assert C.sizeof(F.RtCamera) == 8 + 6 * 24 + 8 and C.sizeof(F.RtGridDesc) == 4 * 24 + 8
    assert C.sizeof(F.RtStats) == 10 * 8 + 5 * 8


def test_no_gpu_means_a_loud_error_not_a_fallback():
    """The product has no CPU path: without a device rt_init fails and every entry point refuses to run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = F.hip_lib()
    assert lib.rt_init(0) != 0
    assert b"no CPU fallback" in lib.rt_last_error() or b"HIP" in lib.rt_last_error()
    assert lib.rt_scene_create(None, 0, None, 0) is None
    st = F.RtStats()
    assert lib.rt_render(None, None, 1, 0, 1, 1, 1, None, C.byref(st)) != 0


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(F, "LIB_DIR", str(tmp_path))
    monkeypatch.setattr(F, "_hip", None)
    with pytest.raises(ImportError, match="no CPU fallback"):
        F.hip_lib()


BUILD_CASES = [(s, m) for s in ("simple_fuzzy_metal", "simple_hollow_glass", "triangle_test", "triangle_materials",
                                "weekend-1-50-25", "weekend-5-80-15", "weekend-final") for m in ("sah-position", "sah-plane", "naive")]
BUILD_CASES += [("dragon_res4", "sah-plane"), ("dragon_res4", "naive"), ("mixed", "sah-position")]


@pytest.mark.parametrize("scene,mode", BUILD_CASES)
def test_bvh_builder_is_bit_identical_to_the_literal_restatement(scene, mode):
    """The host's sweep builder vs the oracle's line-by-line O(N^2) Bvh::new (bvh.rs:19-180): nodes and object order."""
    w = weekend_mixed() if scene == "mixed" else load_world(scene)
    w.algorithm, w.bvh_mode = "bvh", mode
    w.init()
    nodes, order = oracle_of(w).build_bvh(mode).get_bvh()
    assert nodes.tobytes() == w.bvh.nodes().tobytes()
    assert np.array_equal(order, w.bvh.order(len(w)))
    # shape invariants the device relies on
    n = w.bvh.nodes()
    inner = n[n["is_leaf"] == 0]
    assert np.all(inner["right"] == inner["left"] + 1)
    assert n[n["is_leaf"] == 1]["count"].sum() == len(w)


def test_bvh_builder_full_sah_on_res4_dragon():
    """11 102 triangles, full SAH sweep: ~15 s for the literal oracle builder, so it runs once here."""
    w = load_world("dragon_res4"); w.algorithm = "bvh"; w.init()
    nodes, order = oracle_of(w).build_bvh("sah-position").get_bvh()
    assert nodes.tobytes() == w.bvh.nodes().tobytes() and np.array_equal(order, w.bvh.order(len(w)))


@pytest.mark.parametrize("scene,size", [("simple_fuzzy_metal", 25), ("simple_fuzzy_metal", 5), ("simple_fuzzy_metal", 1), ("triangle_materials", 5),
                                        ("weekend-final", 25), ("weekend-final", 5), ("dragon_res4", 25), ("dragon_res4", 5), ("mixed", 5)])
def test_grid_builder_is_bit_identical_to_the_literal_restatement(scene, size):
    w = weekend_mixed() if scene == "mixed" else load_world(scene)
    w.algorithm, w.grid_size = "grid", size
    w.init()
    aabb, off, pl = oracle_of(w).build_grid(size).get_grid()
    assert np.array_equal(aabb, w.grid.cell_aabb()) and np.array_equal(off, w.grid.cell_offset()) and np.array_equal(pl, w.grid.prim_list())


def test_grid_sizes_the_reference_panics_on_are_errors():
    from raytracer_amd import RtError
    w = load_world("triangle_test"); w.algorithm = "grid"
    for bad in (0.5, 3.0):                      # step_by(0) panics; 3 does not tile 200 -> index out of bounds
        w.grid_size = bad
        with pytest.raises(RtError):
            w.init()


def test_ragged_and_empty_inputs():
    from raytracer_amd import RtError
    with pytest.raises(RtError):
        HittableList.from_json("scenes/does_not_exist.json")
    with pytest.raises(RtError):
        HittableList.from_ply("scenes/simple_fuzzy_metal.json")          # not a PLY
    e = HittableList.from_arrays(np.zeros(0, dtype=[("type", "<u4"), ("material", "<u4"), ("v", "<f8", (9,))]), np.zeros((0, 3)),
                                 np.zeros(0, dtype=[("type", "<u4"), ("pad", "<u4"), ("albedo", "<f8", (3,)), ("param", "<f8")]))
    e.algorithm = "bvh"
    with pytest.raises(RtError):
        e.init()                                                         # reference: nodes.first() is None, every ray misses


def test_subdivision_keeps_winding_and_area():
    w = load_world("triangle_test")
    before = w.prims()
    F.host_lib().rth_scene_subdivide(w._h, 2)
    after = w.prims()
    assert len(after) == 16 * len(before)
    def normal(v): return np.cross(v[3:6] - v[0:3], v[6:9] - v[0:3])
    for i, p in enumerate(before):
        n0 = normal(p["v"]); acc = np.zeros(3)
        for q in after[16 * i:16 * i + 16]:
            n = normal(q["v"]); assert np.dot(n, n0) > 0; acc += n
        assert np.allclose(acc, n0, rtol=1e-12)


def _outname(*argv):
    args = [b"raytracer"] + [a.encode() for a in argv]
    buf = C.create_string_buffer(512)
    rc = F.host_lib().rth_output_filename(len(args), (C.c_char_p * len(args))(*args), buf, 512)
    return rc, buf.value.decode()


def test_cli_output_filenames_follow_the_reference():
    """get_output_filename (rtweekend.rs:175-197) + Display for Options (rtweekend.rs:97-113)."""
    assert _outname("scenes/weekend-final.json") == (0, "output/weekend-final-bvh.ppm")
    assert _outname("scenes/weekend-final.json", "--algorithm", "naive") == (0, "output/weekend-final-naive.ppm")
    assert _outname("scenes/weekend-final.json", "-o", "bvh-sah-plane") == (0, "output/weekend-final-bvh-BvhSahPlane.ppm")
    assert _outname("scenes/x.json", "--algorithm", "grid") == (0, "output/x-grid-size=25.ppm")
    assert _outname("scenes/x.json", "--algorithm=grid", "-g", "2.5") == (0, "output/x-grid-size=2.5.ppm")
    assert _outname("scenes/dragon_recon/dragon_vrip_res4.ply", "--format", "ply", "--camera", "3", "-o", "bvh-naive") == \
        (0, "output/dragon_vrip_res4-bvh-BvhNaive_pos3.ppm")
    assert _outname("a.json", "-o", "bvh-naive", "-o", "bvh-sah-plane")[0] == -2          # check_valid_options rtweekend.rs:150-160
    assert _outname("a.json", "--algorithm", "octree")[0] == -1
    assert _outname("a.json", "--bogus")[0] == -1
    assert _outname("a.json", "--gpus", "8")[0] == 0 and _outname("a.json", "--gpus", "0")[0] == -1 and _outname("a.json", "--gpus=65")[0] == -1


def test_camera_presets_follow_main_rs():
    """main.rs:72-104: the preset is chosen by the *relative* prefix of the file name."""
    c = Camera("scenes/weekend-final.json")
    assert (c.image_width, c.image_height, c.samples_per_pixel, c.max_depth) == (900, 506, 50, 50)
    assert list(c.c.center) == [13.0, 2.0, 3.0]
    assert list(Camera("scenes/dragon_recon/dragon_vrip_res4.ply").c.center) == [0.0, 15.0, 50.0]
    assert list(Camera("scenes/dragon_recon/dragon_vrip_res4.ply", camera=2).c.center) == [80.0, 15.0, 10.0]
    assert list(Camera("/abs/scenes/weekend-final.json").c.center) == [0.0, 0.0, 0.0]
    assert (Camera("scenes/weekend-final.json", image_width=1920).image_height, Camera("x", image_width=3840).image_height) == (1080, 2160)
    from raytracer_amd import RtError
    with pytest.raises(RtError):
        Camera("scenes/dragon_recon/x.ply", camera=7)                                      # "Camera position does not exist"


def test_camera_matches_oracle_initialise():
    from oracle import oracle_py as O
    for name, kw in (("scenes/weekend-final.json", dict(vfov=20.0, look_from=(13, 2, 3), look_at=(0, 0, 0), focus_dist=1.0)),
                     ("scenes/dragon_recon/a.ply", dict(vfov=20.0, look_from=(0, 15, 50), look_at=(0, 12, 0), focus_dist=50.0)),
                     ("other.json", dict(vfov=90.0, look_from=(0, 0, 0), look_at=(0, 0, -1), focus_dist=1.0))):
        for width in (900, 64, 1920):
            a = Camera(name, image_width=width).c
            b = O.camera(16.0 / 9.0, width, kw["vfov"], kw["look_from"], kw["look_at"], (0, 1, 0), 0.1, kw["focus_dist"])
            assert bytes(a) == bytes(b)


def test_resolve_matches_oracle():
    from oracle import oracle_py as O
    rng = np.random.default_rng(1)
    s = rng.uniform(-0.5, 80.0, (40, 30, 3))
    assert np.array_equal(Camera("x", image_width=64).resolve(s, 50), O.resolve(s, 50))


def test_cli_binary_help_and_errors():
    exe = "raytracer_amd/bin/raytracer"
    if not os.path.exists(exe):
        pytest.skip("CLI not built")
    r = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--algorithm" in r.stdout and "--grid-size" in r.stdout and "--reduce" in r.stdout
    r = subprocess.run([exe, "scenes/simple_fuzzy_metal.json", "-o", "bvh-naive", "-o", "bvh-sah-plane"], capture_output=True, text=True)
    assert r.returncode != 0 and "Can't have the following options at the same time" in r.stderr
    r = subprocess.run([exe, "scenes/simple_fuzzy_metal.json", "--stats"], capture_output=True, text=True)
    assert r.returncode == 0 and "scenes/simple_fuzzy_metal.json & 4 & 2 & 2 & 0" in r.stdout          # main.rs:107-116


@pytest.mark.parametrize("mode", ["sah-position", "sah-plane", "naive"])
def test_parallel_bvh_build_is_identical_to_the_sequential_one(mode, monkeypatch):
    """Big nodes are split with helper threads and their subtrees built concurrently, then merged by index remapping
    (accel.cpp: build_par); RT_BVH_PAR_MIN lowers the node size at which that happens so that a small mesh exercises it."""
    def build(par_min):
        monkeypatch.setenv("RT_BVH_PAR_MIN", str(par_min))
        w = load_world("dragon_res4"); w.algorithm, w.bvh_mode = "bvh", mode; w.init()
        return w.bvh.nodes().tobytes(), w.bvh.order(len(w)).tobytes(), w.bvh.depth
    seq = build(10 ** 9)
    for par_min in (64, 1000):
        assert build(par_min) == seq


def test_ply_float_parser_is_correctly_rounded(tmp_path):
    """parse_f32 (scene.cpp) must equal strtof bit for bit: random decimals, short and long mantissas, exact f32 midpoints."""
    import subprocess
    exe = str(tmp_path / "parse_f32_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Iinclude", "-o", exe, "tests/native/parse_f32_check.cpp"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-500:]


def _write_ascii_ply(path, verts, faces, extra_vertex_props=0, crlf=False):
    nl = "\r\n" if crlf else "\n"
    with open(path, "w", newline="") as f:
        f.write("ply" + nl + "format ascii 1.0" + nl + f"element vertex {len(verts)}" + nl)
        f.write("property float x" + nl + "property float y" + nl + "property float z" + nl)
        for k in range(extra_vertex_props):
            f.write(f"property float extra{k}" + nl)
        f.write(f"element face {len(faces)}" + nl + "property list uchar int vertex_indices" + nl + "end_header" + nl)
        tail = "".join(" 0.5" for _ in range(extra_vertex_props))
        f.write(nl.join(f"{np.float32(x):.9g} {np.float32(y):.9g}  {np.float32(z):.9g}{tail} " for x, y, z in verts) + nl)
        f.write(nl.join(f"3 {a} {b} {c} " for a, b, c in faces) + nl)


def test_parallel_ply_parse_equals_the_token_parser(tmp_path, monkeypatch):
    """big ASCII elements are parsed by several threads, rows split at line starts; the result must be
    the token-by-token parser's, and anything that is not one-row-per-line falls back to it (same errors)."""
    rng = np.random.default_rng(9)
    nv, nf = 5000, 9000
    verts = (rng.normal(size=(nv, 3)) * [0.1, 0.07, 0.05] + [0, 0.12, 0]).astype(np.float32)
    verts[:7] = [[0, 0, 0], [1e-30, -1e-30, 3.4e38], [0.1, 0.2, 0.3], [-0.0, 1, 2], [1.17549435e-38, 5e-39, 1e-45], [16777217, 0.30000001192092896, 1e10], [7, 8, 9]]
    faces = rng.integers(0, nv, size=(nf, 3))

    def load(path, par):
        monkeypatch.setenv("RT_PLY_PAR_MIN", "0" if par else "18446744073709551615")
        w = HittableList.from_ply(str(path))
        return w.prims().tobytes(), w.centroids().tobytes()

    for extra, crlf in ((0, False), (2, False), (1, True)):
        path = tmp_path / f"m{extra}{int(crlf)}.ply"
        _write_ascii_ply(path, verts, faces, extra, crlf)
        monkeypatch.setenv("RT_PLY_THREADS", "5")
        assert load(path, True) == load(path, False)
    # the reference's own mesh
    ref = "scenes/dragon_recon/dragon_vrip_res4.ply"
    assert load(ref, True) == load(ref, False)
    # rows that span lines, a short face row and an out-of-range index: the thread path declines, the token parser decides
    good = open(tmp_path / "m00.ply").read()
    head, body = good.split("end_header\n")
    lines = body.split("\n")
    spanning = head + "end_header\n" + "\n".join([lines[0].replace(" ", "\n", 1)] + lines[1:])
    (tmp_path / "span.ply").write_text(spanning)
    assert load(tmp_path / "span.ply", True) == load(tmp_path / "m00.ply", False)
    bad_idx = head + "end_header\n" + "\n".join(lines[:nv] + [f"3 0 1 {nv}"] + lines[nv + 1:])
    (tmp_path / "bad.ply").write_text(bad_idx)
    msgs = []
    for par in (True, False):
        with pytest.raises(RtError) as e:
            load(tmp_path / "bad.ply", par)
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1] and "not found" in msgs[0]


def test_parallel_ply_parse_time_on_a_full_size_mesh(tmp_path, monkeypatch):
    """A 765 k-triangle ASCII PLY (the bench mesh written out; the reference's full dragon is 871 k): both parsers agree; times printed."""
    import time
    w = HittableList.from_ply("scenes/dragon_recon/dragon_vrip_res3.ply", subdivide=2)
    tri = w.prims()["v"].reshape(-1, 3, 3) / 100.0
    verts, inv = np.unique(tri.reshape(-1, 3), axis=0, return_inverse=True)
    faces = inv.reshape(-1, 3)
    path = tmp_path / "big.ply"
    with open(path, "w") as f:
        f.write(f"ply\nformat ascii 1.0\nelement vertex {len(verts)}\nproperty float x\nproperty float y\nproperty float z\n"
                f"element face {len(faces)}\nproperty list uchar int vertex_indices\nend_header\n")
        np.savetxt(f, verts.astype(np.float32), fmt="%.9g")
        np.savetxt(f, np.hstack([np.full((len(faces), 1), 3), faces]), fmt="%d")
    out = {}
    for par in (False, True):
        monkeypatch.setenv("RT_PLY_PAR_MIN", "0" if par else "18446744073709551615")
        t = time.time(); m = HittableList.from_ply(str(path)); dt = time.time() - t
        out[par] = (m.prims().tobytes(), dt)
    assert out[True][0] == out[False][0] and len(out[True][0]) == len(faces) * 80
    assert np.array_equal(np.frombuffer(out[True][0], dtype=w.prims().dtype)["v"], w.prims()["v"])       # and it is the mesh that was written
    print(f"\nASCII PLY, {len(faces)} faces, {os.path.getsize(path) >> 20} MiB: token parser {out[False][1]:.2f} s, threads {out[True][1]:.2f} s")


def test_in_process_run_refuses_gpus_flag():
    """`--gpus N` forks workers before the first HIP call, which only the CLI binary can guarantee: rth_run (in-process) says so."""
    import raytracer_amd as rt
    from raytracer_amd.host import RtError
    with pytest.raises(RtError, match="raytracer binary"):
        rt.run(["scenes/simple_fuzzy_metal.json", "--gpus", "2"])
    with pytest.raises(RtError, match="raytracer binary"):              # one worker, but a forked one
        rt.run(["scenes/simple_fuzzy_metal.json", "--gpus", "1", "--reduce", "rccl"])
    with pytest.raises(RtError, match="possible values: auto, rccl, host"):
        rt.run(["scenes/simple_fuzzy_metal.json", "--reduce", "nccl"])
