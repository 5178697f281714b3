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


def test_noalloc_allocates_nothing(gpu):
    """The property the entry point is named for: free device memory (hipMemGetInfo, through torch) is the same before and after
    the call, on a scene whose own workspace has never been created — so neither a hipMalloc'd workspace, nor a grown event
    pool, nor anything else that takes device memory can hide in it.  The first call of a process loads the kernels' code
    object (HIP does that lazily), so one warm-up call on another scene comes first."""
    warm = weekend_mixed(); warm.algorithm = "bvh"; warm.init().upload()  # same kind of scene: the same kernel instantiations
    fb = torch.zeros((270, 480, 3), dtype=torch.float64, device="cuda")
    Camera(warm.filename, image_width=64, samples_per_pixel=1).render(warm, d_out=fb.data_ptr(), noalloc=True)
    w = weekend_mixed(); w.algorithm = "bvh"; w.init().upload()          # fresh scene: rt_render_device has never run on it
    cam = Camera(w.filename, image_width=480, samples_per_pixel=3)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    _, s1 = cam.render(w, seed=3, d_out=fb.data_ptr(), noalloc=True)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    _, s2 = Camera(w.filename, image_width=96, samples_per_pixel=2).render(w, seed=4, d_out=fb.data_ptr(), noalloc=True)
    torch.cuda.synchronize()
    free2 = torch.cuda.mem_get_info()[0]
    assert free1 == free0 and free2 == free0, f"free device memory moved across rt_render_device_noalloc: {free0} -> {free1} -> {free2}"
    assert s1["primary_rays"] == 480 * 270 * 3 and s2["primary_rays"] == 96 * 54 * 2
    # the allocating entry point on the same scene does take memory (the check above can see a workspace)
    cam.render(w, seed=3, d_out=fb.data_ptr())
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] < free0
    w.close(); warm.close()


def test_noalloc_depth_zero_and_no_samples(gpu):
    """camera.rs:155-157: max_depth 0 is black before world.hit is called; s_count 0 is an empty sum.  Both leave through the
    memset path of the bounce loop, which the static workspace must take as well."""
    w = load_world("simple_fuzzy_metal"); w.algorithm = "bvh"; w.init().upload()
    for kw in ({"max_depth": 0, "samples_per_pixel": 2}, {"samples_per_pixel": 2}):
        cam = Camera(w.filename, image_width=64, **kw)
        fa = torch.full((cam.image_height, 64, 3), -1.0, dtype=torch.float64, device="cuda"); fb = fa.clone()
        s_count = None if "max_depth" in kw else 0
        _, sa = cam.render(w, seed=9, s_count=s_count, d_out=fa.data_ptr())
        _, sb = cam.render(w, seed=9, s_count=s_count, d_out=fb.data_ptr(), noalloc=True)
        torch.cuda.synchronize()
        assert torch.equal(fa, fb) and float(fb.abs().sum()) == 0.0
        for k in KEYS:
            assert sa[k] == sb[k], (k, sa[k], sb[k])
    w.close()


def test_noalloc_on_a_callers_stream(gpu):
    """A non-NULL stream argument: the call's kernels are ordered on the caller's stream, after the work already queued there."""
    w = load_world("triangle_materials"); w.algorithm = "bvh"; w.init().upload()
    cam = Camera(w.filename, image_width=160, samples_per_pixel=2)
    ref = torch.zeros((cam.image_height, 160, 3), dtype=torch.float64, device="cuda")
    _, sa = cam.render(w, seed=12, d_out=ref.data_ptr())
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        out = torch.full_like(ref, 7.0)                       # queued on s: the render must come after this fill
        _, sb = cam.render(w, seed=12, d_out=out.data_ptr(), stream=s.cuda_stream, noalloc=True)
        doubled = out * 2.0                                   # and this after the render
    s.synchronize()
    assert torch.equal(out, ref) and torch.equal(doubled, ref * 2.0)
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
