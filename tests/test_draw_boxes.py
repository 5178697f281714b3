"""`-o draw-boxes`: the sticky rec.hits_aabb_edge of aabb.rs:57-60,90-97 and the red returns of
ray_color (camera.rs:162-164,178-180).  CPU: hand-derived known answers for the oracle.  GPU: the HIP path against the oracle —
edge flags, hit results and counters bit-exact, radiance within RADIANCE_RTOL."""
import numpy as np
import pytest

from oracle import oracle_py as O

from .util import bits, load_world, random_rays, setup_pair

ONE_SPHERE = (np.array([(0, 0, (0, 0, -5, 1, 0, 0, 0, 0, 0))], dtype=O.PRIM_DT), np.array([[0.0, 0.0, -5.0]]),
              np.array([(0, 0, (0.5, 0.5, 0.5), 0.0)], dtype=O.MAT_DT))


def one_sphere_oracle(draw):
    osc = O.OracleScene(*ONE_SPHERE)
    osc.build_bvh("sah-position")            # one leaf: the root box is [-1,1] x [-1,1] x [-6,-4]
    osc.set_draw_boxes(draw)
    return osc


def test_oracle_edge_flag_known_answers():
    o = np.zeros((4, 3))
    d = np.array([[0.0, 0.0, -1.0],       # enters the front face at its centre (0,0,-4): on one face only -> no edge
                  [1.0, 1.0, -4.0],       # enters at the corner (1,1,-4): three faces -> edge; misses the sphere
                  [1.0, 0.0, -4.0],       # enters at (1,0,-4): x and z faces -> edge; misses the sphere
                  [0.0, 5.0, -1.0]])      # misses the box: is_edge is only evaluated on a successful box test
    r = one_sphere_oracle(True).trace("bvh", o, d)
    assert r["hit"].tolist() == [True, False, False, False]
    assert r["edge"].tolist() == [False, True, True, False]
    off = one_sphere_oracle(False).trace("bvh", o, d)
    assert not off["edge"].any() and off["hit"].tolist() == r["hit"].tolist() and np.array_equal(bits(off["t"]), bits(r["t"]))


def test_oracle_edge_tolerance_is_one_hundredth_inclusive():
    # entry point (x, 0.995, -4): y within 0.01 of max.y = 1 and z on the front face -> edge; y = 0.98 -> not
    o = np.zeros((2, 3)); d = np.array([[0.0, 0.995, -4.0], [0.0, 0.98, -4.0]])
    assert one_sphere_oracle(True).trace("bvh", o, d)["edge"].tolist() == [True, False]


def test_oracle_ray_color_returns_red_for_hit_and_for_miss():
    cam = O.camera(16.0 / 9.0, 64, 90.0, (0, 0, 0), (0, 0, -1), (0, 1, 0), 0.0, 1.0)
    plain, _ = one_sphere_oracle(False).render(cam, "bvh", 0, 4, 50, seed=3)
    drawn, c = one_sphere_oracle(True).render(cam, "bvh", 0, 4, 50, seed=3)
    changed = np.any(plain != drawn, axis=-1)
    assert changed.any() and not changed.all()
    # a sample that touched an edge contributes exactly (1,0,0) at depth 0 (the throughput is still 1): G and B only lose light
    assert (drawn[..., 1] <= plain[..., 1] + 1e-12).all() and (drawn[..., 2] <= plain[..., 2] + 1e-12).all()
    assert c["primary_rays"] == 64 * 36 * 4


# ---------------------------------------------------------------------------------------------------------------- GPU
GPU_CASES = [("simple_fuzzy_metal", "bvh", "sah-position", None), ("simple_fuzzy_metal", "grid", None, 5),
             ("triangle_materials", "bvh", "sah-plane", None), ("weekend-5-80-15", "bvh", "sah-position", None),
             ("weekend-5-80-15", "grid", None, 5), ("dragon_res4", "bvh", "sah-plane", None), ("dragon_res4", "grid", None, 5)]


@pytest.mark.gpu
@pytest.mark.parametrize("scene,algorithm,bvh_mode,grid_size", GPU_CASES)
def test_gpu_edge_flags_and_render_match_oracle(gpu, scene, algorithm, bvh_mode, grid_size):
    from raytracer_amd import Camera
    from .test_gpu_parity import RADIANCE_RTOL, assert_trace_equal, primary_rays
    w = load_world(scene)
    w.draw_boxes = True
    osc = setup_pair(w, algorithm, bvh_mode or "sah-position", grid_size or 25.0)
    osc.set_draw_boxes(True)
    w.upload()
    cam = Camera(w.filename, image_width=160)
    o, d = cam.get_rays(*primary_rays(cam), sample=0, seed=7)
    for what, (ro, rd) in (("primary", (o, d)), ("random", random_rays(20000, 11, -15.0, 25.0))):
        g, c = w.hit(ro, rd), osc.trace(algorithm, ro, rd)
        assert np.array_equal(g["edge"], c["edge"]), f"{what}: edge flags differ at {np.flatnonzero(g['edge'] != c['edge'])[:10]}"
        assert_trace_equal(g, c, f"{scene}/{algorithm} draw-boxes {what}")
    assert w.hit(o, d)["edge"].any(), "the overlay marks nothing on this view"
    cam = Camera(w.filename, image_width=96, samples_per_pixel=4)
    img, st = cam.render(w, seed=5)
    ref, cst = osc.render(O.camera_from_rt(cam.c), algorithm, 0, 4, cam.max_depth, seed=5, nthreads=8)
    for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb", "gridbox_intersection_checks"):
        assert st[k] == cst[k], f"counter {k}: gpu {st[k]} oracle {cst[k]}"
    rel = np.max(np.abs(img - ref) / np.maximum(np.abs(ref), 1e-300))
    assert rel <= RADIANCE_RTOL, f"max relative radiance error {rel}"
    # switching the option off again restores the plain path (the flag is per scene, not sticky in the library)
    w.draw_boxes = False; w.upload(); osc.set_draw_boxes(False)
    assert not w.hit(o, d)["edge"].any()
    assert_trace_equal(w.hit(o, d), osc.trace(algorithm, o, d), "option off")
    w.close()
