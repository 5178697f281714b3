"""BASELINE.json configs[1], [2] and [4] at their FULL sizes, driver-run (`-m gpu`): weekend-final 1920x1080x256, the res4 dragon
900x506x50 through the BVH and through the grid (size 5), and the mixed sphere/triangle scene at 900x506x4096 (the "compaction
stress": several batches, path ids up to 1.9e9).  The oracle cannot render these in seconds, so the checks are the
size-independent properties of the path (tests/test_gpu_parity.py holds the same scenes to the oracle bit for bit at
64-320 px):

  * `primary_rays` is exactly W*H*spp (camera.rs:70) and every other counter is an integer that ADDS over a split of the
    sample range: counters([0,k)) == counters([0,j)) + counters([j,k)) — every path is a pure function of (seed, pixel, global
    sample index), whatever call, batch or queue position it runs in;
  * the radiance sums are finite, non-negative and at most spp per channel (every attenuation and the sky are <= 1);
  * sum([0,k)) equals sum([0,j)) + sum([j,k)) up to the re-association of one f64 addition chain per pixel (relative 1e-12;
    a call adds its samples in ascending order, so only the split point re-associates);
  * a render forced into a different number of batches (RT_BATCH_PATHS) is BIT-identical: batches continue the same ordered sum.
"""
import numpy as np
import pytest
import torch

from raytracer_amd import Camera

from .util import load_world, weekend_mixed

pytestmark = pytest.mark.gpu
COUNTERS = ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb", "gridbox_intersection_checks",
            "ray_segments", "box_tests", "prim_tests")


def render_dev(cam, w, s_begin, s_count, seed=1):
    fb = torch.empty((cam.image_height, cam.image_width, 3), dtype=torch.float64, device="cuda")
    _, st = cam.render(w, seed=seed, s_begin=s_begin, s_count=s_count, d_out=fb.data_ptr())
    torch.cuda.synchronize()
    return fb, st


def check_split(cam, w, spp, j, label):
    whole, st = render_dev(cam, w, 0, spp)
    assert st["primary_rays"] == cam.image_width * cam.image_height * spp, label
    assert bool(torch.isfinite(whole).all()) and float(whole.min()) >= 0.0 and float(whole.max()) <= spp + 1e-9, label
    a, sa = render_dev(cam, w, 0, j)
    b, sb = render_dev(cam, w, j, spp - j)
    for k in COUNTERS:
        assert sa[k] + sb[k] == st[k], (label, k, sa[k], sb[k], st[k])
    err = float(((a + b) - whole).abs().max()); scale = float(whole.abs().max())
    assert err <= 1e-12 * max(scale, 1.0), (label, err, scale)
    return whole, st


def with_batch_paths(monkeypatch, rt, paths, fn):
    try:
        monkeypatch.setenv("RT_BATCH_PATHS", str(paths)); rt.init(0)
        return fn()
    finally:
        monkeypatch.delenv("RT_BATCH_PATHS", raising=False); rt.init(0)


def test_config1_weekend_final_1920x1080x256(gpu, monkeypatch):
    w = load_world("weekend-final"); w.algorithm = "bvh"; w.init().upload()
    cam = Camera(w.filename, image_width=1920, samples_per_pixel=256)
    assert (cam.image_width, cam.image_height, cam.max_depth) == (1920, 1080, 50)
    whole, st = check_split(cam, w, 256, 100, "weekend-final")
    assert st["scatter_rays"] > st["primary_rays"]                        # 487 spheres fill the frame: more than one bounce per path
    # the same 256 samples in batches of 37 and 219... : 1920*1080*40 paths per batch -> 7 batches
    forced, sf = with_batch_paths(monkeypatch, gpu, 1920 * 1080 * 40, lambda: render_dev(cam, w, 0, 256))
    assert torch.equal(forced, whole), "batching changed the image"
    for k in COUNTERS:
        assert sf[k] == st[k], k
    w.close()


@pytest.mark.parametrize("algorithm,grid_size", [("bvh", None), ("grid", 5.0)])
def test_config2_dragon_res4_900x506x50(gpu, monkeypatch, algorithm, grid_size):
    w = load_world("dragon_res4"); w.algorithm = algorithm
    if grid_size:
        w.grid_size = grid_size
    w.init().upload()
    cam = Camera(w.filename, image_width=900, samples_per_pixel=50)
    assert (cam.image_width, cam.image_height, cam.max_depth) == (900, 506, 50)
    whole, st = check_split(cam, w, 50, 17, f"res4/{algorithm}")
    if algorithm == "grid":
        assert st["gridbox_intersection_checks"] > 0 and st["overlapping_aabb"] == 0
    else:
        assert st["overlapping_aabb"] > 0 and st["gridbox_intersection_checks"] == 0
    forced, sf = with_batch_paths(monkeypatch, gpu, 900 * 506 * 8, lambda: render_dev(cam, w, 0, 50))
    assert torch.equal(forced, whole), "batching changed the image"
    for k in COUNTERS:
        assert sf[k] == st[k], k
    w.close()


def test_config4_mixed_900x506x4096(gpu, monkeypatch):
    """1.87e9 paths: more than one batch even with all of HBM (196 B of queue state per path), path ids close to 2^31 per
    batch.  A forced four-batch run (1024 samples each) must give the same bits as the default batching."""
    w = weekend_mixed(); w.algorithm = "bvh"; w.init().upload()
    cam = Camera(w.filename, image_width=900, samples_per_pixel=4096)
    assert (cam.image_width, cam.image_height) == (900, 506)
    whole, st = render_dev(cam, w, 0, 4096)
    assert st["primary_rays"] == 900 * 506 * 4096
    assert bool(torch.isfinite(whole).all()) and float(whole.min()) >= 0.0 and float(whole.max()) <= 4096 + 1e-9
    forced, sf = with_batch_paths(monkeypatch, gpu, 900 * 506 * 1024, lambda: render_dev(cam, w, 0, 4096))
    assert torch.equal(forced, whole), "a four-batch run differs from the default batching"
    for k in COUNTERS:
        assert sf[k] == st[k], k
    # counters add over a split of the sample range (one half each; sums re-associate at the split point only)
    a, sa = render_dev(cam, w, 0, 2048)
    b, sb = render_dev(cam, w, 2048, 2048)
    for k in COUNTERS:
        assert sa[k] + sb[k] == st[k], (k, sa[k], sb[k], st[k])
    assert float(((a + b) - whole).abs().max()) <= 1e-12 * float(whole.abs().max())
    w.close()
