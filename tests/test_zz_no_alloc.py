"""rt_render_device_noalloc (include/rt_hip.h): the render entry point for callers that cannot allocate.  Same arguments as
rt_render_device, workspace in a static device array; checked bit for bit against rt_render_device on two calls in a row,
the second with a smaller input, and for its documented limits."""
import numpy as np
import pytest
import torch

from raytracer_amd import Camera
from raytracer_amd.host import RtError

from .util import load_world, weekend_mixed

pytestmark = pytest.mark.gpu
KEYS = ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb", "gridbox_intersection_checks", "ray_segments")


def _both(w, cam, seed, s_begin=0):
    fa = torch.full((cam.image_height, cam.image_width, 3), -1.0, dtype=torch.float64, device="cuda")
    fb = torch.full((cam.image_height, cam.image_width, 3), -2.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    _, sa = cam.render(w, seed=seed, s_begin=s_begin, d_out=fa.data_ptr())
    _, sb = cam.render(w, seed=seed, s_begin=s_begin, d_out=fb.data_ptr(), noalloc=True)
    torch.cuda.synchronize()
    return fa.cpu().numpy(), sa, fb.cpu().numpy(), sb


def test_noalloc_equals_render_device_twice_second_call_smaller(gpu):
    w = weekend_mixed(); w.algorithm = "bvh"; w.init().upload()
    # 1st call: 480x270 = 129 600 pixels, 5 spp -> batches of 2 samples fit the 262 144-path workspace (3 batches)
    cam = Camera(w.filename, image_width=480, samples_per_pixel=5)
    a, sa, b, sb = _both(w, cam, seed=21)
    assert np.array_equal(a, b), "first call differs from rt_render_device"
    for k in KEYS:
        assert sa[k] == sb[k], (k, sa[k], sb[k])
    # 2nd call, smaller image and another sample range: nothing left over from the first call may show
    cam2 = Camera(w.filename, image_width=96, samples_per_pixel=3, max_depth=7)
    a2, sa2, b2, sb2 = _both(w, cam2, seed=22, s_begin=4)
    assert np.array_equal(a2, b2), "second (smaller) call differs from rt_render_device"
    for k in KEYS:
        assert sa2[k] == sb2[k], (k, sa2[k], sb2[k])
    w.close()


@pytest.mark.parametrize("scene,algorithm,grid_size", [("dragon_res4", "bvh", None), ("dragon_res4", "grid", 5.0), ("simple_hollow_glass", "naive", None)])
def test_noalloc_other_walks(gpu, scene, algorithm, grid_size):
    w = load_world(scene); w.algorithm = algorithm
    if grid_size:
        w.grid_size = grid_size
    w.init().upload()
    cam = Camera(w.filename, image_width=128, samples_per_pixel=2)
    a, sa, b, sb = _both(w, cam, seed=5)
    assert np.array_equal(a, b)
    for k in KEYS:
        assert sa[k] == sb[k], (k, sa[k], sb[k])
    w.close()


def test_noalloc_limits_are_errors(gpu):
    w = load_world("simple_fuzzy_metal"); w.algorithm = "bvh"; w.init().upload()
    fb = torch.zeros((540, 960, 3), dtype=torch.float64, device="cuda")
    with pytest.raises(RtError, match="static workspace"):        # 960x540 = 518 400 pixels > 262 144
        Camera(w.filename, image_width=960, samples_per_pixel=1).render(w, d_out=fb.data_ptr(), noalloc=True)
    with pytest.raises(RtError, match="max_depth"):
        Camera(w.filename, image_width=64, samples_per_pixel=1, max_depth=2000).render(w, d_out=fb.data_ptr(), noalloc=True)
    w.close()
