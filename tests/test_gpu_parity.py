"""GPU parity tests: the HIP path (through the C ABI of include/rt_hip.h) against the CPU oracle.

Bar (task ③): bit-exact for hit decisions — primitive index, rec.t bits, the t of rec.p, the boolean, and every
Data counter; radiance sums within a stated relative tolerance because the device multiplies attenuations
left-to-right where the reference nests them right-to-left (camera.rs:172).
"""
import numpy as np
import pytest

from oracle import oracle_py as O
from raytracer_amd import Camera

from .util import bits, load_world, oracle_of, random_rays, setup_pair, weekend_mixed

pytestmark = pytest.mark.gpu

RADIANCE_RTOL = 1e-12   # per-pixel sums; a product of <= 50 factors re-associated is within ~50 ulp


def assert_trace_equal(g, c, what):
    assert np.array_equal(g["hit"], c["hit"]), f"{what}: hit flags differ at {np.flatnonzero(g['hit'] != c['hit'])[:10]}"
    assert np.array_equal(g["prim"], c["prim"]), f"{what}: primitive indices differ at {np.flatnonzero(g['prim'] != c['prim'])[:10]}"
    assert np.array_equal(bits(g["t"]), bits(c["t"])), f"{what}: rec.t bits differ"
    m = c["prim"] >= 0
    assert np.array_equal(bits(g["tp"][m]), bits(c["tp"][m])), f"{what}: rec.p parameter differs"
    for k in ("intersection_checks", "traversal_steps", "overlapping_aabb", "gridbox_intersection_checks", "box_tests", "prim_tests"):
        assert g["stats"][k] == c["stats"][k], f"{what}: counter {k}: gpu {g['stats'][k]} oracle {c['stats'][k]}"


def primary_rays(cam, step=1, sample=0, seed=1):
    jj, ii = np.meshgrid(np.arange(0, cam.image_height, step), np.arange(0, cam.image_width, step), indexing="ij")
    return ii.ravel().astype(np.uint32), jj.ravel().astype(np.uint32)


CASES = [
    # scene, algorithm, bvh_mode, grid_size, image width
    ("simple_fuzzy_metal", "naive", None, None, 160),
    ("simple_fuzzy_metal", "bvh", "sah-position", None, 160),
    ("simple_fuzzy_metal", "bvh", "sah-plane", None, 160),
    ("simple_fuzzy_metal", "bvh", "naive", None, 160),
    ("simple_fuzzy_metal", "grid", None, 25, 160),
    ("simple_fuzzy_metal", "grid", None, 5, 160),
    ("simple_fuzzy_metal", "grid", None, 1, 96),
    ("simple_hollow_glass", "bvh", "sah-position", None, 160),
    ("triangle_test", "bvh", "sah-position", None, 160),
    ("triangle_materials", "naive", None, None, 160),
    ("triangle_materials", "bvh", "sah-position", None, 160),
    ("triangle_materials", "bvh", "sah-plane", None, 160),
    ("triangle_materials", "bvh", "naive", None, 160),
    ("triangle_materials", "grid", None, 25, 160),
    ("triangle_materials", "grid", None, 5, 160),
    ("weekend-5-80-15", "naive", None, None, 160),
    ("weekend-5-80-15", "bvh", "sah-position", None, 160),
    ("weekend-5-80-15", "grid", None, 5, 160),
    ("weekend-final", "bvh", "sah-position", None, 320),
    ("weekend-final", "bvh", "sah-plane", None, 160),
    ("weekend-final", "bvh", "naive", None, 160),
    ("weekend-final", "grid", None, 25, 96),
    ("weekend-final", "grid", None, 5, 160),
    ("dragon_res4", "bvh", "sah-position", None, 320),
    ("dragon_res4", "bvh", "sah-plane", None, 160),
    ("dragon_res4", "bvh", "naive", None, 160),
    ("dragon_res4", "grid", None, 25, 64),
    ("dragon_res4", "grid", None, 5, 160),
]


@pytest.mark.parametrize("scene,algorithm,bvh_mode,grid_size,width", CASES)
def test_primary_and_random_hits_bit_exact(gpu, scene, algorithm, bvh_mode, grid_size, width):
    """north_star: primary-ray hit primitive indices bit-exact against Hittable::hit on the same ray batch."""
    w = load_world(scene)
    big = scene.startswith("dragon") and algorithm == "bvh" and bvh_mode == "sah-position"
    osc = setup_pair(w, algorithm, bvh_mode or "sah-position", grid_size or 25.0, oracle_own_build=not big)
    w.upload()
    cam = Camera(w.filename, image_width=width)
    pi, pj = primary_rays(cam)
    o, d = cam.get_rays(pi, pj, sample=0, seed=7)
    oo, od = O.generate(O.camera_from_rt(cam.c), 7, 0, pi, pj)
    assert np.array_equal(bits(o), bits(oo)) and np.array_equal(bits(d), bits(od)), "get_ray differs from the oracle"
    assert_trace_equal(w.hit(o, d), osc.trace(algorithm, o, d), f"{scene}/{algorithm} primary")
    # incoherent rays from inside and outside the scene bounds, both interval ends finite
    n_rand = 20000 if algorithm != "naive" or len(w) < 200 else 4000
    ro, rd = random_rays(n_rand, 11, -15.0, 25.0)
    assert_trace_equal(w.hit(ro, rd, 0.001, 1e9), osc.trace(algorithm, ro, rd, 0.001, 1e9), f"{scene}/{algorithm} random")
    w.close()


def test_bvh_layout_consumed_matches_oracle_build(gpu):
    """north_star: the flattened BVH node layout is bit-exact against the reference's SAH build (res4, literal O(N^2) oracle)."""
    w = load_world("dragon_res4")
    osc = setup_pair(w, "bvh", "sah-position", oracle_own_build=True)
    nodes, order = osc.get_bvh()
    assert nodes.tobytes() == w.bvh.nodes().tobytes()
    assert np.array_equal(order, w.bvh.order(len(w)))
    w.upload()
    ro, rd = random_rays(50000, 3, -15.0, 25.0)
    assert_trace_equal(w.hit(ro, rd), osc.trace("bvh", ro, rd), "res4 own-build")
    w.close()


RENDER_CASES = [
    ("simple_fuzzy_metal", "bvh", None, 96, 8, 50),
    ("simple_fuzzy_metal", "naive", None, 96, 4, 50),
    ("simple_fuzzy_metal", "grid", 5, 64, 4, 50),
    ("simple_hollow_glass", "bvh", None, 96, 8, 50),
    ("simple_shiny_metal", "bvh", None, 96, 4, 50),
    ("triangle_materials", "bvh", None, 96, 8, 50),
    ("triangle_materials", "grid", 25, 64, 4, 50),
    ("weekend-final", "bvh", None, 128, 4, 50),
    ("weekend-final", "bvh", None, 64, 2, 3),
    ("dragon_res4", "bvh", None, 128, 4, 50),
    ("dragon_res4", "grid", 5, 48, 2, 50),
    ("mixed", "bvh", None, 96, 4, 50),
]


@pytest.mark.parametrize("scene,algorithm,grid_size,width,spp,max_depth", RENDER_CASES)
def test_render_matches_oracle(gpu, scene, algorithm, grid_size, width, spp, max_depth):
    """Same seed, same counter RNG: every path takes the same decisions, so the counters are equal and the
    per-pixel radiance sums agree to RADIANCE_RTOL; the 8-bit image is then identical up to rounding ties."""
    w = weekend_mixed() if scene == "mixed" else load_world(scene)
    big = scene.startswith("dragon") and algorithm == "bvh"
    osc = setup_pair(w, algorithm, "sah-position", grid_size or 25.0, oracle_own_build=not big)
    w.upload()
    cam = Camera(w.filename, image_width=width, samples_per_pixel=spp, max_depth=max_depth)
    img, st = cam.render(w, seed=5)
    ref, cst = osc.render(O.camera_from_rt(cam.c), algorithm, 0, spp, max_depth, seed=5, nthreads=8)
    for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb", "gridbox_intersection_checks"):
        assert st[k] == cst[k], f"counter {k}: gpu {st[k]} oracle {cst[k]}"
    assert st["ray_segments"] <= st["primary_rays"] + st["scatter_rays"]
    scale = np.maximum(np.abs(ref), 1e-300)
    rel = np.max(np.abs(img - ref) / scale)
    assert rel <= RADIANCE_RTOL, f"max relative radiance error {rel}"
    a, b = cam.resolve(img), O.resolve(ref, spp)
    assert np.mean(a != b) <= 1e-4 and np.max(np.abs(a.astype(int) - b.astype(int))) <= 1
    w.close()


def test_spp_split_is_independent_of_the_split(gpu):
    """§8e: samples [0,8) rendered as 8, as 3+5, and one at a time give the same sums up to f64 association."""
    w = load_world("triangle_materials"); w.algorithm = "bvh"; w.init(); w.upload()
    cam = Camera(w.filename, image_width=96, samples_per_pixel=8)
    full, st = cam.render(w, seed=9)
    a, sa = cam.render(w, seed=9, s_begin=0, s_count=3)
    b, sb = cam.render(w, seed=9, s_begin=3, s_count=5)
    assert sa["scatter_rays"] + sb["scatter_rays"] == st["scatter_rays"]
    assert np.allclose(a + b, full, rtol=1e-13, atol=0)
    acc = np.zeros_like(full)
    for s in range(8):
        one, _ = cam.render(w, seed=9, s_begin=s, s_count=1)
        acc = acc + one
    assert np.array_equal(acc, full), "sample-ordered accumulation must be bit-identical"
    w.close()


def test_render_batching_does_not_change_the_image(gpu, monkeypatch):
    w = load_world("simple_hollow_glass"); w.algorithm = "bvh"; w.init(); w.upload()
    cam = Camera(w.filename, image_width=64, samples_per_pixel=6)
    ref, _ = cam.render(w, seed=2)
    one_by_one = np.zeros_like(ref)
    for s in range(6):
        img, _ = cam.render(w, seed=2, s_begin=s, s_count=1)
        one_by_one += img
    assert np.array_equal(ref, one_by_one)
    w.close()


@pytest.mark.parametrize("scene,width,spp", [("simple_hollow_glass", 70, 13), ("dragon_res4", 61, 64), ("weekend-5-80-15", 48, 100)])
def test_queue_order_does_not_change_the_image(gpu, monkeypatch, scene, width, spp):
    """k_generate may put 1, 4, 16 or 64 samples of a pixel into one wave (RT_LG_SAMPLES) and the streaming kernels may run
    on any grid (RT_STREAM_BLOCKS): paths land in the queues in a different order, every path and the ordered per-pixel
    sum stay what they were, so the framebuffer and the counters are bit-identical.  Odd sizes: partial tiles and groups."""
    import raytracer_amd as rt
    w = load_world(scene); w.algorithm = "bvh"; w.init(); w.upload()
    cam = Camera(w.filename, image_width=width, samples_per_pixel=spp)
    ref = None
    try:
        for lg, blocks in (("7", "4096"), ("0", "4096"), ("2", "64"), ("4", "4096"), ("6", "8")):
            monkeypatch.setenv("RT_LG_SAMPLES", lg); monkeypatch.setenv("RT_STREAM_BLOCKS", blocks); rt.init(0)
            img, st = cam.render(w, seed=5)
            key = (img.tobytes(), st["scatter_rays"], st["intersection_checks"], st["traversal_steps"], st["overlapping_aabb"])
            if ref is None: ref = key
            assert key == ref, f"RT_LG_SAMPLES={lg} RT_STREAM_BLOCKS={blocks}"
    finally:
        monkeypatch.setenv("RT_LG_SAMPLES", "7"); monkeypatch.setenv("RT_STREAM_BLOCKS", "4096"); rt.init(0)     # the library's defaults
    w.close()


def test_reserve_allocates_ahead_and_changes_nothing(gpu):
    """rt_scene_reserve sizes the workspace a render would; the image is the one an unreserved scene renders."""
    from raytracer_amd import RtError
    a = load_world("simple_hollow_glass"); a.algorithm = "bvh"; a.init(); a.upload()
    b = load_world("simple_hollow_glass"); b.algorithm = "bvh"; b.init(); b.upload()
    cam = Camera(a.filename, image_width=80, samples_per_pixel=5)
    cam.reserve(b); cam.reserve(b, 2)                                  # twice, and for fewer samples: no-ops after the first
    ia, sa = cam.render(a, seed=3); ib, sb = cam.render(b, seed=3)
    assert np.array_equal(ia, ib) and sa["scatter_rays"] == sb["scatter_rays"]
    big = Camera(a.filename, image_width=200, samples_per_pixel=9)       # a larger render afterwards still resizes by itself
    assert np.array_equal(big.render(a, seed=3)[0], big.render(b, seed=3)[0])
    c = load_world("simple_hollow_glass")
    with pytest.raises(RtError):
        cam.reserve(c)                                                 # not uploaded
    a.close(); b.close()


def test_full_size_properties_weekend_final(gpu):
    """configs[1] at full size (1920x1080; 2 of the 256 spp): size-independent properties only —
    exact primary count, finite non-negative radiance bounded by the sky colour, and the image of samples
    [0,2) equals image[0,1) + image[1,2)."""
    w = load_world("weekend-final"); w.algorithm = "bvh"; w.init(); w.upload()
    cam = Camera(w.filename, image_width=1920, samples_per_pixel=2)
    assert (cam.image_width, cam.image_height) == (1920, 1080)
    img, st = cam.render(w, seed=1)
    assert st["primary_rays"] == 1920 * 1080 * 2
    assert np.isfinite(img).all() and img.min() >= 0.0 and img.max() <= 2.0 * 1.0 + 1e-9
    a, _ = cam.render(w, seed=1, s_begin=0, s_count=1)
    b, _ = cam.render(w, seed=1, s_begin=1, s_count=1)
    assert np.array_equal(a + b, img)
    w.close()


def test_errors_are_reported_not_crashed(gpu):
    from raytracer_amd import RtError
    w = load_world("triangle_test")
    w.upload()
    with pytest.raises(RtError):
        w.hit(np.zeros((1, 3)), np.ones((1, 3)), algorithm="bvh")      # reference: "Please run HittableList.init() first"
    with pytest.raises(RtError):
        w.hit(np.zeros((1, 3)), np.ones((1, 3)), algorithm="grid")
    assert w.hit(np.zeros((0, 3)), np.zeros((0, 3)), algorithm="naive")["prim"].shape == (0,)
    with pytest.raises(RtError, match="no grid"):
        w.device_grid()                                                  # rt_scene_get_grid before any grid was installed
    with pytest.raises(RtError, match="grid size"):
        w.build_grid_on_device(0.5)                                      # grid.rs:36 step_by(0) panics in the reference
    with pytest.raises(RtError):
        w.build_grid_on_device(float("nan"))
    w.build_grid_on_device(25.0)
    assert w.device_grid()[0].n_cells == 512 and w.hit(np.zeros((1, 3)), np.ones((1, 3)), algorithm="grid")["prim"].shape == (1,)
    w.close()


def test_reference_form_extend_kernel_agrees(gpu, monkeypatch):
    """RT_EXTEND_VARIANT=1 selects the plain one-ray-per-thread extend kernel (kept for A/B runs): same bits."""
    import raytracer_amd as rt
    w = load_world("dragon_res4")
    osc = setup_pair(w, "bvh", "sah-plane")
    w.upload()
    ro, rd = random_rays(30000, 5, -15.0, 25.0)
    ref = osc.trace("bvh", ro, rd)
    try:
        monkeypatch.setenv("RT_EXTEND_VARIANT", "1"); rt.init(0)
        assert_trace_equal(w.hit(ro, rd), ref, "variant 1")
    finally:
        monkeypatch.setenv("RT_EXTEND_VARIANT", "2"); rt.init(0)
    assert_trace_equal(w.hit(ro, rd), ref, "variant 2")
    w.close()


def test_compact_lattice_layout_is_chosen_for_ply_and_is_bit_identical(gpu, monkeypatch):
    """PLY coordinates lie on the (f64)f32*100 lattice (parser.rs:67-73), so rt_scene_set_bvh picks the 64-B pair /
    48-B triangle forms; RT_COMPACT=0 forces the f64 forms.  Same bits either way, and both equal the oracle."""
    import raytracer_amd as rt
    w = load_world("dragon_res4")
    osc = setup_pair(w, "bvh", "sah-position", oracle_own_build=False)
    ro, rd = random_rays(40000, 21, -15.0, 25.0)
    ref = osc.trace("bvh", ro, rd)
    try:
        monkeypatch.setenv("RT_COMPACT", "0"); rt.init(0); w.upload()
        assert w.bvh_layout() == (112, 80)
        plain = w.hit(ro, rd)
    finally:
        monkeypatch.setenv("RT_COMPACT", "1"); rt.init(0)
    w.upload()
    assert w.bvh_layout() == (64, 48)
    compact = w.hit(ro, rd)
    assert_trace_equal(plain, ref, "f64 layout")
    assert_trace_equal(compact, ref, "compact layout")
    # the opt-in 32-byte "new planes only" sibling record (DPairN, RT_PAIRN=1): two accesses per visit, same bits, same counters
    try:
        monkeypatch.setenv("RT_PAIRN", "1"); rt.init(0); w.upload()
        assert w.bvh_layout() == (32, 48)
        pn = w.hit(ro, rd)
        cam = Camera(w.filename, image_width=160, samples_per_pixel=3)
        img_n, st_n = cam.render(w, seed=4)
    finally:
        monkeypatch.delenv("RT_PAIRN"); rt.init(0)
    w.upload()
    img_c, st_c = cam.render(w, seed=4)
    assert_trace_equal(pn, ref, "new-planes layout")
    assert np.array_equal(img_n, img_c)
    for k in ("scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb", "ray_segments"):
        assert st_n[k] == st_c[k], k
    w.close()
    # JSON scenes carry arbitrary f64 coordinates: no lattice; f64 pairs (112 B), 48-B spheres when there is nothing else
    j = load_world("weekend-5-80-15"); j.algorithm = "bvh"; j.init(); j.upload()
    assert j.bvh_layout() == (112, 48)
    j.close()
    m = weekend_mixed(); m.algorithm = "bvh"; m.init(); m.upload()
    assert m.bvh_layout() == (112, 80)
    m.close()


def test_subdivided_ply_stays_on_the_lattice_and_matches_oracle(gpu):
    """The synthetic stand-in for the absent full dragon (bench.py): res4 split twice; midpoints are rounded onto the
    PLY lattice so the compact layout applies; hits are bit-exact against the oracle on the same mesh."""
    from raytracer_amd import HittableList
    w = HittableList.from_ply("scenes/dragon_recon/dragon_vrip_res4.ply", subdivide=1)
    assert len(w) == 4 * 11102
    v = w.prims()["v"]
    f = (v / 100.0).astype(np.float32)
    assert np.array_equal(f.astype(np.float64) * 100.0, v), "subdivision left the f32*100 lattice"
    osc = setup_pair(w, "bvh", "sah-position", oracle_own_build=False)
    w.upload()
    assert w.bvh_layout() == (64, 48)
    cam = Camera(w.filename, image_width=256)
    o, d = cam.get_rays(*primary_rays(cam), sample=0, seed=3)
    assert_trace_equal(w.hit(o, d), osc.trace("bvh", o, d), "subdivided res4 primary")
    w.close()


def test_lattice_scale_one_and_f32_exact_json_like_scene(gpu):
    """Coordinates that are f32-exact but not multiples of the PLY scale select the S = 1 lattice."""
    from raytracer_amd import HittableList
    rng = np.random.default_rng(4)
    n = 300
    v = rng.uniform(-8, 8, size=(n, 9)).astype(np.float32).astype(np.float64)
    v[:, 3:6] = (v[:, 0:3] + rng.uniform(-1, 1, size=(n, 3))).astype(np.float32)
    v[:, 6:9] = (v[:, 0:3] + rng.uniform(-1, 1, size=(n, 3))).astype(np.float32)
    prims = np.zeros(n, dtype=O.PRIM_DT); prims["type"] = 1; prims["v"] = v
    cent = (v[:, 0:3] + v[:, 3:6] + v[:, 6:9]) / 3.0
    mats = np.zeros(1, dtype=O.MAT_DT); mats["albedo"] = 0.5
    w = HittableList.from_arrays(prims, cent, mats, "scenes/none.json")
    osc = setup_pair(w, "bvh", "sah-position")
    w.upload()
    assert w.bvh_layout() == (64, 48)
    ro, rd = random_rays(20000, 8, -10.0, 10.0)
    assert_trace_equal(w.hit(ro, rd), osc.trace("bvh", ro, rd), "S=1 lattice")
    w.close()


def test_full_size_properties_dragon_4k(gpu, monkeypatch):
    """configs[3] at full size (764 704-triangle stand-in mesh, 3840x2160; 2 of the 128 spp per GPU): the oracle cannot run
    this in seconds, so size-independent properties — exact primary count; image of samples [0,2) == [0,1) + [1,2) bit for
    bit; and the compact f32-lattice layout renders the same bits and counters as the f64 layout (RT_COMPACT=0)."""
    import raytracer_amd as rt
    from raytracer_amd import HittableList
    w = HittableList.from_ply("scenes/dragon_recon/dragon_vrip_res3.ply", subdivide=2)
    w.algorithm = "bvh"; w.gpu_build = True; w.init(); w.upload()
    assert len(w) == 764704 and w.bvh_layout() == (64, 48)
    cam = Camera(w.filename, image_width=3840, samples_per_pixel=2)
    assert (cam.image_width, cam.image_height) == (3840, 2160)
    img, st = cam.render(w, seed=1)
    assert st["primary_rays"] == 3840 * 2160 * 2
    assert np.isfinite(img).all() and img.min() >= 0.0 and img.max() <= 2.0 + 1e-9
    a, sa = cam.render(w, seed=1, s_begin=0, s_count=1)
    b, sb = cam.render(w, seed=1, s_begin=1, s_count=1)
    assert np.array_equal(a + b, img) and sa["scatter_rays"] + sb["scatter_rays"] == st["scatter_rays"]
    try:
        monkeypatch.setenv("RT_COMPACT", "0"); rt.init(0); w.upload()
        assert w.bvh_layout() == (112, 80)
        plain, sp = cam.render(w, seed=1)
    finally:
        monkeypatch.setenv("RT_COMPACT", "1"); rt.init(0)
    assert np.array_equal(plain, img)
    for k in ("scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb"):
        assert sp[k] == st[k], k
    w.close()


@pytest.mark.parametrize("max_depth", [0, 1, 2])
def test_shallow_depths_match_oracle(gpu, max_depth):
    """camera.rs:155-157: depth 0 returns black before world.hit is called (no counters but primary_rays); depth 1 and 2 end
    every path after one / two scatters."""
    w = weekend_mixed()
    osc = setup_pair(w, "bvh")
    w.upload()
    cam = Camera(w.filename, image_width=96, samples_per_pixel=3, max_depth=max_depth)
    img, st = cam.render(w, seed=9)
    ref, cst = osc.render(O.camera_from_rt(cam.c), "bvh", 0, 3, max_depth, seed=9, nthreads=4)
    for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb"):
        assert st[k] == cst[k], f"depth {max_depth}, counter {k}: gpu {st[k]} oracle {cst[k]}"
    if max_depth == 0:
        assert not img.any() and not ref.any() and st["scatter_rays"] == 0 and st["intersection_checks"] == 0
    else:
        assert np.max(np.abs(img - ref) / np.maximum(np.abs(ref), 1e-300)) <= RADIANCE_RTOL
    w.close()


def test_max_depth_limit_is_an_error_not_a_fault(gpu):
    from raytracer_amd.host import RtError
    w = load_world("simple_fuzzy_metal"); w.algorithm = "bvh"; w.init().upload()
    cam = Camera(w.filename, image_width=32, samples_per_pixel=1, max_depth=0xFFFFFFFE)
    with pytest.raises(RtError, match="max_depth"):
        cam.render(w)
    w.close()


def test_set_bvh_rejects_trees_the_pair_tables_cannot_address(gpu):
    """rt_scene_set_bvh validates what the kernels trust: sibling adjacency AND the odd index of every left child
    (bvh.rs:167-169 numbers children 2k+1, 2k+2; the persistent kernel's pair tables are addressed by (left+1)/2)."""
    import ctypes as C
    from raytracer_amd import _ffi as F
    w = load_world("simple_fuzzy_metal"); w.algorithm = "bvh"; w.init().upload()
    nodes, order = w.bvh.nodes(), w.bvh.order(len(w))
    assert not nodes[0]["is_leaf"] and nodes[0]["left"] == 1
    # the same tree with an unused node spliced in at index 1: children of the root become (2, 3) — adjacent, but left is even
    bad = np.zeros(len(nodes) + 1, dtype=nodes.dtype)
    bad[0] = nodes[0]; bad[2:] = nodes[1:]; bad[1] = nodes[1]
    for nd in bad:
        if not nd["is_leaf"]:
            nd["left"] += 1; nd["right"] += 1
    hip = F.hip_lib()
    rc = hip.rt_scene_set_bvh(w._dev, bad.ctypes.data_as(C.POINTER(F.RtBvhNode)), len(bad), order.ctypes.data_as(C.POINTER(C.c_uint32)), len(order))
    assert rc == -1 and b"even index" in hip.rt_last_error()
    # the scene is still usable with the good tree
    rc = hip.rt_scene_set_bvh(w._dev, nodes.ctypes.data_as(C.POINTER(F.RtBvhNode)), len(nodes), order.ctypes.data_as(C.POINTER(C.c_uint32)), len(order))
    assert rc == 0
    w.close()


def test_extend_blocks_raised_after_set_bvh_stays_inside_the_stack(gpu, monkeypatch):
    """The continuation stack is sized when the tree is installed; a later rt_init with a larger RT_EXTEND_BLOCKS must not
    launch more blocks than it was sized for: same image, same counters."""
    import raytracer_amd as rt
    w = load_world("dragon_res4")
    monkeypatch.setenv("RT_EXTEND_BLOCKS", "256"); rt.init(0)
    w.algorithm = "bvh"; w.init().upload()
    cam = Camera(w.filename, image_width=160, samples_per_pixel=2)
    a, sa = cam.render(w, seed=2)
    monkeypatch.setenv("RT_EXTEND_BLOCKS", "4096"); rt.init(0)
    b, sb = cam.render(w, seed=2)
    monkeypatch.setenv("RT_EXTEND_BLOCKS", "2048"); rt.init(0)
    assert np.array_equal(a, b) and sa["intersection_checks"] == sb["intersection_checks"] and sa["scatter_rays"] == sb["scatter_rays"]
    w.close()


def _edge_rays(w, n, seed):
    """Rays built to reach the corners of the reference's arithmetic: direction components that are +0 / -0 (1/d = +-inf, and
    0 * inf = NaN wherever the origin lies ON a box plane, aabb.rs:34-51), origins exactly on vertices, box planes and sphere
    centres, axis-aligned and all-zero directions, denormal and huge components, and a few non-finite origins."""
    rng = np.random.default_rng(seed)
    p = w.prims()
    pts = np.concatenate([p["v"][:, 0:3], p["v"][p["type"] == 1][:, 3:6], p["v"][p["type"] == 1][:, 6:9]])
    if w.bvh is not None:
        nd = w.bvh.nodes()
        pts = np.concatenate([pts, nd["min"], nd["max"]])
    o = pts[rng.integers(len(pts), size=n)].copy()
    # mix coordinates of different points so that an origin shares ONE plane with a box and is off the others
    mix = rng.random((n, 3)) < 0.5
    o[mix] = pts[rng.integers(len(pts), size=n)][mix]
    o += rng.choice([0.0, 0.0, 0.0, 1e-12, -3.0, 7.0], size=(n, 1)) * rng.normal(size=(n, 3))
    d = rng.normal(size=(n, 3))
    z = rng.random((n, 3)) < 0.35
    d[z] = rng.choice([0.0, -0.0], size=int(z.sum()))
    k = n // 16
    d[:k] *= 1e-310                                             # denormal components: 1/d overflows to inf
    d[k:2 * k] *= 1e200
    o[2 * k:2 * k + 8, 0] = [np.inf, -np.inf, np.nan, 1e308, -1e308, 0.0, -0.0, 5e-324]
    return o, d


@pytest.mark.parametrize("scene,algorithm,grid_size", [("triangle_materials", "bvh", None), ("triangle_materials", "naive", None),
                                                       ("triangle_materials", "grid", 5.0), ("weekend-5-80-15", "bvh", None),
                                                       ("weekend-5-80-15", "grid", 25.0), ("mixed", "bvh", None), ("mixed", "grid", 5.0),
                                                       ("dragon_res4", "bvh", None), ("dragon_res4", "grid", 5.0), ("dragon_res4", "grid", 1.0)])
def test_degenerate_rays_bit_exact(gpu, scene, algorithm, grid_size):
    """NaN / inf / signed-zero behaviour of the slab, sphere and triangle tests and of the grid's saturating casts is the
    reference's, comparison by comparison: hit flags, primitives, rec.t / rec.p bits and all counters equal the oracle's."""
    w = weekend_mixed() if scene == "mixed" else load_world(scene)
    osc = setup_pair(w, algorithm, "sah-position", grid_size or 25.0, oracle_own_build=scene != "dragon_res4")
    w.upload()
    o, d = _edge_rays(w, 6000 if algorithm != "naive" else 2000, 31)
    with np.errstate(all="ignore"):
        assert_trace_equal(w.hit(o, d), osc.trace(algorithm, o, d), f"{scene}/{algorithm} degenerate rays, tmax inf")
        assert_trace_equal(w.hit(o, d, 0.001, 50.0), osc.trace(algorithm, o, d, 0.001, 50.0), f"{scene}/{algorithm} degenerate rays, tmax 50")
    w.close()


def _random_scene(seed, n=60):
    """A scene the reference's loaders could produce but none of the committed files does: overlapping spheres and triangles,
    zero-radius spheres, zero-area and sliver triangles, coincident primitives, metal fuzz 0 / 1, refraction indices around 1."""
    from raytracer_amd import HittableList
    from raytracer_amd.host import _MAT_DT, _PRIM_DT
    rng = np.random.default_rng(seed)
    prims = np.zeros(n, _PRIM_DT)
    prims["type"] = rng.integers(0, 2, size=n)
    c = rng.uniform(-4, 4, size=(n, 3)) + np.array([0.0, 0.0, -6.0])
    for i in range(n):
        if prims["type"][i] == 0:
            prims["v"][i, :3] = c[i]; prims["v"][i, 3] = rng.choice([0.0, 0.05, 0.5, 1.5, 3.0])
        else:
            tri = c[i] + rng.normal(scale=rng.choice([1e-9, 0.3, 2.0]), size=(3, 3))
            if rng.random() < 0.1: tri[2] = tri[1]                       # zero area
            prims["v"][i] = tri.ravel()
    prims[n - 1] = prims[0]; prims[n - 2] = prims[1]                     # coincident primitives: order decides
    mats = np.zeros(8, _MAT_DT)
    mats["type"] = [0, 0, 1, 1, 1, 2, 2, 2]
    mats["albedo"] = rng.uniform(0.1, 1.0, size=(8, 3))
    mats["param"] = [0, 0, 0.0, 1.0, 0.3, 1.5, 1.0, 0.7]
    prims["material"] = rng.integers(0, 8, size=n)
    v = prims["v"]
    cent = np.where((prims["type"] == 0)[:, None], v[:, :3], ((v[:, 0:3] + v[:, 3:6]) + v[:, 6:9]) / 3.0)
    return HittableList.from_arrays(prims, cent, mats, filename="scenes/random.json")


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("algorithm,grid_size", [("bvh", None), ("grid", 5.0), ("naive", None)])
def test_random_scenes_render_matches_oracle(gpu, seed, algorithm, grid_size):
    w = _random_scene(seed)
    osc = setup_pair(w, algorithm, "sah-position", grid_size or 25.0)
    w.upload()
    cam = Camera(w.filename, image_width=64, samples_per_pixel=6, max_depth=12)
    img, st = cam.render(w, seed=seed)
    ref, cst = osc.render(O.camera_from_rt(cam.c), algorithm, 0, 6, 12, seed=seed, nthreads=8)
    for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb", "gridbox_intersection_checks"):
        assert st[k] == cst[k], f"seed {seed} {algorithm}: counter {k}: gpu {st[k]} oracle {cst[k]}"
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(img), fin)
    assert np.max(np.abs(img[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-300), initial=0.0) <= RADIANCE_RTOL
    w.close()


def _triangle_soup(seed, n, spread, stacked, center=(0.0, 0.0, -6.0)):
    """All-triangle scene for the cooperative grid test (k_extend_grid_coop): `n` triangles crowded into a few grid cells so that
    their lists are chunked (>= 64 entries).  `stacked` adds runs of parallel, equally wound triangles one behind the other: a ray
    through a run is ACCEPTED several times within one 64-entry chunk (closest_so_far changes mid-chunk, grid.rs:288-291), in
    ascending and in descending order of t — and zero-area, coincident and back-facing triangles sit in between."""
    from raytracer_amd import HittableList
    from raytracer_amd.host import _MAT_DT, _PRIM_DT
    rng = np.random.default_rng(seed)
    prims = np.zeros(n, _PRIM_DT); prims["type"] = 1
    ctr = np.array(center)
    c = rng.uniform(-spread, spread, size=(n, 3)) + ctr
    tri = c[:, None, :] + rng.normal(scale=rng.choice([1e-9, 0.4, 1.5, 4.0], size=(n, 1, 1)), size=(n, 3, 3))
    flat = rng.random(n) < 0.05
    tri[flat, 2] = tri[flat, 1]                                          # zero area
    for s in range(stacked):                                             # runs of 12 parallel triangles facing the camera side
        i0 = int(rng.integers(0, n - 12)); base = rng.normal(scale=min(2.5, 2.0 * spread), size=(3, 3)) + ctr
        order = np.arange(12) if s % 2 == 0 else np.arange(12)[::-1]
        for j, dz in enumerate(order):
            tri[i0 + j] = base + np.array([0.0, 0.0, -0.15 * dz])
        if s % 3 == 0:
            tri[i0 + 5] = tri[i0 + 4]                                    # coincident pair: order decides
    prims["v"] = tri.reshape(n, 9)
    mats = np.zeros(3, _MAT_DT); mats["type"] = [0, 1, 2]; mats["albedo"] = rng.uniform(0.2, 1.0, size=(3, 3)); mats["param"] = [0, 0.4, 1.5]
    prims["material"] = rng.integers(0, 3, size=n)
    cent = ((tri[:, 0] + tri[:, 1]) + tri[:, 2]) / 3.0
    return HittableList.from_arrays(prims, cent, mats, filename="scenes/soup.json")


@pytest.mark.parametrize("seed,n,spread,grid_size,center", [(1, 64, 0.8, 25.0, (10.0, 10.0, -12.0)), (2, 65, 0.8, 25.0, (10.0, 10.0, -12.0)),
                                                            (3, 128, 1.0, 25.0, (10.0, 10.0, -12.0)), (4, 700, 3.0, 25.0, (0.0, 0.0, -6.0)),
                                                            (5, 1500, 6.0, 5.0, (0.0, 0.0, -6.0)), (6, 3000, 9.0, 5.0, (0.0, 0.0, -6.0)),
                                                            (7, 129, 0.5, 5.0, (2.5, 2.5, -7.5))])
def test_cooperative_grid_cells_bit_exact(gpu, monkeypatch, seed, n, spread, grid_size, center):
    """The wave-cooperative cell test against the oracle's literal GridBox::hit: list lengths on both sides of the chunk size,
    several acceptances inside one chunk, clobbered rec.t / rec.p from failed tests (triangle.rs:52,59), rays of a wave in different
    cells (groups of one) and in the same cell (groups of 64), degenerate rays; traces, counters and a rendered image.
    RT_GRID_COOP_SHARE=0 / RT_GRID_COOP_MIN=64 make every grid with a cell of >= 64 triangles take the cooperative kernel (the
    defaults want lists of >= 128 holding half of all entries)."""
    import raytracer_amd as rt
    w = _triangle_soup(seed, n, spread, stacked=max(2, n // 60), center=center)
    osc = setup_pair(w, "grid", "sah-position", grid_size)
    ln = np.diff(w.grid.cell_offset().astype(np.int64))
    assert (ln >= 64).any(), "no cell of this scene is chunked: the test would not reach the cooperative path"
    if seed in (1, 2):
        assert ln.max() == n, "the whole soup was meant to sit in one cell"
    try:
        monkeypatch.setenv("RT_GRID_COOP_SHARE", "0"); monkeypatch.setenv("RT_GRID_COOP_MIN", "64"); rt.init(0)
        w.upload()
    finally:
        monkeypatch.delenv("RT_GRID_COOP_SHARE"); monkeypatch.delenv("RT_GRID_COOP_MIN"); rt.init(0)
    ctr = np.array(center)
    rng = np.random.default_rng(seed)
    # rays through the stacks and the crowd, from both sides; coherent bundles (one origin, a fan of directions) and scattered ones
    o1 = np.tile(ctr + np.array([[0.0, 0.0, 8.0]]), (4096, 1)); d1 = np.array([0.0, 0.0, -1.0]) + rng.normal(scale=0.35, size=(4096, 3))
    o2 = rng.uniform(-8, 8, size=(6000, 3)) + ctr; d2 = rng.normal(size=(6000, 3))
    o3, d3 = _edge_rays(w, 2000, seed + 40)
    o = np.concatenate([o1, o2, o3]); d = np.concatenate([d1, d2, d3])
    with np.errstate(all="ignore"):
        assert_trace_equal(w.hit(o, d), osc.trace("grid", o, d), f"soup {seed}: tmax inf")
        assert_trace_equal(w.hit(o, d, 0.001, 3.0), osc.trace("grid", o, d, 0.001, 3.0), f"soup {seed}: tmax 3")
    cam = Camera(w.filename, image_width=96, samples_per_pixel=4, max_depth=10)
    img, st = cam.render(w, seed=seed)
    ref, cst = osc.render(O.camera_from_rt(cam.c), "grid", 0, 4, 10, seed=seed, nthreads=8)
    for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "gridbox_intersection_checks"):
        assert st[k] == cst[k], f"soup {seed}: counter {k}: gpu {st[k]} oracle {cst[k]}"
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(img), fin)
    assert np.max(np.abs(img[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-300), initial=0.0) <= RADIANCE_RTOL
    w.close()


@pytest.mark.parametrize("scene,bvh_mode", [("weekend-final", "sah-position"), ("weekend-final", "naive"), ("weekend-final", "sah-plane"),
                                            ("weekend-5-80-15", "sah-position"), ("simple_hollow_glass", "sah-position"),
                                            ("mixed", "sah-position"), ("mixed", "naive")])
def test_lds_resident_tree_bit_exact(gpu, monkeypatch, scene, bvh_mode):
    """k_extend_bvh2<0, P, T, LDS>: a small scene's pair and primitive tables staged in LDS by 1024-thread workgroups, four
    continuation-stack entries per lane in LDS (the rest in HBM).  By default only queues of >= 65 536 rays take it;
    RT_LDS_TREE=3 sends every queue through it and fails the call if the scene does not qualify.  Same traces, counters and
    image as the oracle: all-sphere (48-byte records) and mixed (80-byte records) scenes, deep `naive` trees (stack spills),
    degenerate rays, both interval ends finite."""
    import raytracer_amd as rt
    w = weekend_mixed() if scene == "mixed" else load_world(scene)
    osc = setup_pair(w, "bvh", bvh_mode)
    w.upload()
    try:
        monkeypatch.setenv("RT_LDS_TREE", "3"); rt.init(0)
        cam = Camera(w.filename, image_width=200)
        pi, pj = primary_rays(cam)
        o, d = cam.get_rays(pi, pj, sample=0, seed=3)
        assert_trace_equal(w.hit(o, d), osc.trace("bvh", o, d), f"lds {scene}/{bvh_mode} primary")
        ro, rd = random_rays(70000, 19, -15.0, 25.0)
        assert_trace_equal(w.hit(ro, rd, 0.001, 1e9), osc.trace("bvh", ro, rd, 0.001, 1e9), f"lds {scene}/{bvh_mode} random")
        assert_trace_equal(w.hit(ro[:20000], rd[:20000], 0.001, 4.0), osc.trace("bvh", ro[:20000], rd[:20000], 0.001, 4.0), f"lds {scene}/{bvh_mode} tmax 4")
        eo, ed = _edge_rays(w, 4000, 77)
        with np.errstate(all="ignore"):
            assert_trace_equal(w.hit(eo, ed), osc.trace("bvh", eo, ed), f"lds {scene}/{bvh_mode} degenerate")
        cam = Camera(w.filename, image_width=128, samples_per_pixel=4, max_depth=50)
        img, st = cam.render(w, seed=5)
    finally:
        monkeypatch.delenv("RT_LDS_TREE"); rt.init(0)
    ref, cst = osc.render(O.camera_from_rt(cam.c), "bvh", 0, 4, 50, seed=5, nthreads=8)
    for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb"):
        assert st[k] == cst[k], f"counter {k}: gpu {st[k]} oracle {cst[k]}"
    assert np.max(np.abs(img - ref) / np.maximum(np.abs(ref), 1e-300)) <= RADIANCE_RTOL
    # and the default launch rule (this queue length keeps the 128-thread kernel) gives the same bits
    img2, st2 = cam.render(w, seed=5)
    assert np.array_equal(bits(img), bits(img2))
    for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb", "box_tests", "prim_tests", "ray_segments"):
        assert st2[k] == st[k], k
    w.close()


def test_lds_resident_tree_refuses_lattice_scenes(gpu, monkeypatch):
    """RT_LDS_TREE=3 is the tests' way to know which kernel ran: a scene that does not qualify (f32-lattice triangles keep the
    64-byte pairs in HBM) must fail the call, not fall back silently."""
    import raytracer_amd as rt
    w = load_world("dragon_res4"); w.algorithm = "bvh"; w.init().upload()
    ro, rd = random_rays(1000, 1, -15.0, 25.0)
    try:
        monkeypatch.setenv("RT_LDS_TREE", "3"); rt.init(0)
        with pytest.raises(rt.RtError, match="RT_LDS_TREE=3"):
            w.hit(ro, rd)
    finally:
        monkeypatch.delenv("RT_LDS_TREE"); rt.init(0)
    assert w.hit(ro, rd)["hit"].shape == (1000,)
    w.close()


@pytest.mark.parametrize("scene,bvh_mode", [("dragon_res4", "sah-position"), ("dragon_res4", "naive"), ("dragon_res4", "sah-plane"), ("triangle_test", "sah-position")])
def test_line_paired_pair_table_is_bit_identical(gpu, monkeypatch, scene, bvh_mode):
    """The lattice pair table is renumbered on the device side so that a pair's bigger child shares its 128-byte line
    (pair_line_order, rt_api.hip; default for trees of >= 32 768 pairs — the bench mesh and dragon_res3).  RT_PAIR_LINES_MIN=0 applies it
    to the small test meshes too: traces (counters included) and a render equal the oracle's and the reference-order table's
    (RT_PAIR_LINES=0), also combined with the opt-in DPairN records, whose table follows the same order."""
    import raytracer_amd as rt
    w = load_world(scene)
    osc = setup_pair(w, "bvh", bvh_mode, oracle_own_build=(scene != "dragon_res4" or bvh_mode != "sah-position"))
    ro, rd = random_rays(40000, 23, -15.0, 25.0)
    eo, ed = _edge_rays(w, 3000, 5)
    ref = osc.trace("bvh", ro, rd, 0.001, 1e9)
    with np.errstate(all="ignore"):
        ref_e = osc.trace("bvh", eo, ed)
    cam = Camera(w.filename, image_width=120, samples_per_pixel=3)
    imgs = {}
    for label, env in (("reference order", {"RT_PAIR_LINES": "0"}), ("line-paired", {"RT_PAIR_LINES_MIN": "0"}),
                       ("line-paired + DPairN", {"RT_PAIR_LINES_MIN": "0", "RT_PAIRN": "1"})):
        try:
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            rt.init(0); w.upload()
            assert_trace_equal(w.hit(ro, rd, 0.001, 1e9), ref, f"{scene}/{bvh_mode} {label}")
            with np.errstate(all="ignore"):
                assert_trace_equal(w.hit(eo, ed), ref_e, f"{scene}/{bvh_mode} {label} degenerate")
            imgs[label] = cam.render(w, seed=8)
        finally:
            for k in env:
                monkeypatch.delenv(k)
            rt.init(0)
    base_img, base_st = imgs["reference order"]
    for label, (img, st) in imgs.items():
        assert np.array_equal(bits(img), bits(base_img)), label
        for k in ("scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb", "ray_segments"):
            assert st[k] == base_st[k], (label, k)
    w.close()
