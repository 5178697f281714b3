"""The reference's own test matrix (src/test.rs: scenes x {naive, bvh-naive, bvh-sah-plane, bvh-sah-position, grid 25/5/1},
dragon cameras 0..3) re-run as smoke tests through run() of the C++ host — like the reference's tests they assert that the
run completes, plus the side effects the reference documents: a P3 image and one stats.csv row per run."""
import csv
import os

import numpy as np
import pytest

import raytracer_amd as rt

pytestmark = pytest.mark.gpu

J, P = "native", "ply"
MATRIX = []
for scene in ("scenes/simple_fuzzy_metal.json", "scenes/triangle_materials.json", "scenes/weekend-5-80-15.json"):
    MATRIX += [(scene, J, ["--algorithm", "naive"]), (scene, J, ["-o", "bvh-naive"]), (scene, J, ["-o", "bvh-sah-plane"]),
               (scene, J, ["-o", "bvh-sah-position"]), (scene, J, ["--algorithm", "grid", "-g", "25"]),
               (scene, J, ["--algorithm", "grid", "-g", "5"])]
MATRIX += [("scenes/simple_fuzzy_metal.json", J, ["--algorithm", "grid", "-g", "1"])]
for scene, fmt in (("scenes/weekend-final.json", J), ("scenes/dragon_recon/dragon_vrip_res4.ply", P)):
    MATRIX += [(scene, fmt, ["-o", "bvh-naive"]), (scene, fmt, ["-o", "bvh-sah-plane"]), (scene, fmt, ["-o", "bvh-sah-position"]),
               (scene, fmt, ["--algorithm", "grid", "-g", "25"]), (scene, fmt, ["--algorithm", "grid", "-g", "5"])]
MATRIX += [("scenes/dragon_recon/dragon_vrip_res4.ply", P, ["--camera", str(c)]) for c in range(4)]
MATRIX += [("scenes/dragon_recon/dragon_vrip_res4.ply", P, ["--gpu-build"]), ("scenes/weekend-final.json", J, ["--gpu-build"])]


@pytest.mark.parametrize("scene,fmt,extra", MATRIX)
def test_reference_matrix_runs(gpu, tmp_path, scene, fmt, extra):
    out = str(tmp_path / "output")
    argv = [scene, "--format", fmt, "--width", "64", "--spp", "2", "--output-dir", out] + extra
    assert rt.run(argv) == 0
    ppm = [f for f in os.listdir(out) if f.endswith(".ppm")]
    assert len(ppm) == 1
    tok = open(os.path.join(out, ppm[0])).read().split()
    assert tok[:4] == ["P3", "64", "36", "255"] and len(tok) == 4 + 64 * 36 * 3
    px = np.array(tok[4:], dtype=int)
    assert px.min() >= 0 and px.max() <= 255 and px.std() > 0
    rows = list(csv.DictReader(open(os.path.join(out, "stats.csv"))))
    assert len(rows) == 1
    r = rows[0]
    assert list(r.keys()) == ["seconds", "init_time", "primary_rays", "scatter_rays", "intersection_checks", "traversal_steps",
                              "overlapping_aabb", "gridbox_intersection_checks", "filename", "primitives", "algorithm", "options",
                              "image_width", "image_height", "samples_per_pixel", "max_depth"]      # data.rs:6-24
    assert int(r["primary_rays"]) == 64 * 36 * 2 and r["filename"] == scene and int(r["intersection_checks"]) > 0
    assert r["algorithm"] in ("Naive", "BVH", "Grid")


@pytest.mark.parametrize("extra,name", [(["-o", "draw-boxes"], "simple_fuzzy_metal-bvh-DrawBoxes.ppm"),
                                        (["-o", "bvh-sah-plane", "-o", "draw-boxes"], "simple_fuzzy_metal-bvh-BvhSahPlane_DrawBoxes.ppm"),
                                        (["--algorithm", "grid", "-g", "5", "-o", "draw-boxes"], "simple_fuzzy_metal-grid-DrawBoxes_size=5.ppm")])
def test_draw_boxes_overlay(gpu, tmp_path, extra, name):
    """`-o draw-boxes` (rtweekend.rs:146): box edges are painted pure red (camera.rs:162-164,178-180); the option shows in the
    file name (rtweekend.rs:97-113) and combines with a BVH build option (check_valid_options ignores it, rtweekend.rs:150-160)."""
    out = str(tmp_path / "output")
    assert rt.run(["scenes/simple_fuzzy_metal.json", "--width", "160", "--spp", "4", "--output-dir", out] + extra) == 0
    assert os.listdir(out).count(name) == 1
    px = np.array(open(os.path.join(out, name)).read().split()[4:], dtype=int).reshape(-1, 3)
    red = (px[:, 0] == 255) & (px[:, 1] == 0) & (px[:, 2] == 0)
    assert red.sum() > 0, "no pure-red edge pixel in the overlay"
    assert "DrawBoxes" in list(csv.DictReader(open(os.path.join(out, "stats.csv"))))[0]["options"]


def test_binary_ppm_holds_the_same_pixels(gpu, tmp_path):
    """--binary-ppm: P6 bytes == the numbers of the contractual P3 text."""
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    base = ["scenes/simple_hollow_glass.json", "--width", "96", "--spp", "3"]
    assert rt.run(base + ["--output-dir", a]) == 0 and rt.run(base + ["--output-dir", b, "--binary-ppm"]) == 0
    name = "simple_hollow_glass-bvh.ppm"
    p3 = np.array(open(os.path.join(a, name)).read().split()[4:], dtype=np.uint8)
    raw = open(os.path.join(b, name), "rb").read()
    head = b"P6\n96 54\n255\n"
    assert raw.startswith(head) and np.array_equal(np.frombuffer(raw[len(head):], dtype=np.uint8), p3)


def test_gpu_built_grid_gives_the_same_image(gpu, tmp_path):
    """--gpu-build with --algorithm grid (rt_scene_build_grid): same tables, hence the same pixels and counters."""
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    base = ["scenes/weekend-final.json", "--width", "96", "--spp", "2", "--algorithm", "grid", "-g", "5"]
    assert rt.run(base + ["--output-dir", a]) == 0 and rt.run(base + ["--output-dir", b, "--gpu-build"]) == 0
    name = "weekend-final-grid-size=5.ppm"
    assert open(os.path.join(a, name)).read() == open(os.path.join(b, name)).read()
    ra, rb = (list(csv.DictReader(open(os.path.join(d, "stats.csv"))))[0] for d in (a, b))
    for k in ("intersection_checks", "traversal_steps", "gridbox_intersection_checks", "scatter_rays"):
        assert ra[k] == rb[k], k


@pytest.mark.parametrize("opt,name", [("bvh-naive", "weekend-final-bvh-BvhNaive.ppm"), ("bvh-sah-plane", "weekend-final-bvh-BvhSahPlane.ppm")])
def test_gpu_built_bvh_modes_give_the_same_image(gpu, tmp_path, opt, name):
    """--gpu-build with `-o bvh-naive` / `-o bvh-sah-plane` (rt_bvh_build_mode): the same tree as the host builder's, hence the same
    pixels and counters."""
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    base = ["scenes/weekend-final.json", "--width", "96", "--spp", "2", "-o", opt]
    assert rt.run(base + ["--output-dir", a]) == 0 and rt.run(base + ["--output-dir", b, "--gpu-build"]) == 0
    assert open(os.path.join(a, name)).read() == open(os.path.join(b, name)).read()
    ra, rb = (list(csv.DictReader(open(os.path.join(d, "stats.csv"))))[0] for d in (a, b))
    for k in ("intersection_checks", "traversal_steps", "overlapping_aabb", "scatter_rays"):
        assert ra[k] == rb[k], k


def test_cli_gpus_2_equals_gpus_1(gpu, tmp_path):
    """`raytracer --gpus N` (the host adds --gpus): N forked workers, each rendering its spp_range, partial sums
    added in rank order by the parent.  Here N = 2 on the box's one GPU (--device 0 makes every worker use it); the image
    must equal the single-process one (8-bit output: identical up to a rounding tie) and the counters must add up.
    The CLI binary is started as a child process: a forking host must not have touched HIP itself."""
    import subprocess
    exe = os.path.join(os.path.dirname(rt.__file__), "bin", "raytracer")
    outs = []
    for n in (1, 2):
        out = str(tmp_path / f"out{n}")
        cmd = [exe, "scenes/weekend-final.json", "--width", "128", "--spp", "7", "--seed", "4", "--output-dir", out, "--gpus", str(n), "--device", "0"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        px = np.array(open(os.path.join(out, "weekend-final-bvh.ppm")).read().split()[4:], dtype=int)
        row = list(csv.DictReader(open(os.path.join(out, "stats.csv"))))[0]
        outs.append((px, row))
    (p1, r1), (p2, r2) = outs
    for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb"):
        assert r1[k] == r2[k], k
    assert np.max(np.abs(p1 - p2)) <= 1 and np.mean(p1 != p2) < 1e-3
