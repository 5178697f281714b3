"""Grid::new on the GPU (rt_scene_build_grid) against the host builder, which tests/test_host.py pins
against the oracle's literal restatement of grid.rs:23-61: descriptor, cell boxes, offsets and object lists byte for byte."""
import time

import numpy as np
import pytest

from raytracer_amd import HittableList, RtError

from .util import load_world, weekend_mixed

pytestmark = pytest.mark.gpu


def compare(w, size):
    w.algorithm, w.grid_size = "grid", float(size)
    w.gpu_build = False; w.init()
    host = (bytes(w.grid.desc), w.grid.cell_aabb(), w.grid.cell_offset(), w.grid.prim_list())
    w.gpu_build = True; w.init(); t = time.time(); w.upload(); dt = time.time() - t
    d, aabb, off, pl = w.device_grid()
    assert bytes(d) == host[0], "descriptor"
    assert np.array_equal(off, host[2]), f"cell offsets differ at {np.flatnonzero(off != host[2])[:8]}"
    assert len(pl) == len(host[3]) and np.array_equal(pl, host[3]), "object lists"
    assert aabb.tobytes() == host[1].tobytes(), f"cell boxes differ at {np.flatnonzero((aabb != host[1]).any(axis=1))[:8]}"
    return dt


@pytest.mark.parametrize("size", [25, 10, 5, 1])
@pytest.mark.parametrize("scene", ["simple_fuzzy_metal", "triangle_materials", "weekend-final", "mixed", "dragon_res4"])
def test_gpu_grid_equals_host_grid(gpu, scene, size):
    w = weekend_mixed() if scene == "mixed" else load_world(scene)
    compare(w, size)


@pytest.mark.parametrize("size", [3, 7, 1.5, 33.3])
def test_gpu_grid_sizes_that_do_not_tile(gpu, size):
    """Sizes that do not divide 200: the host builder either succeeds (truncated step, unreachable cells) or reports the
    reference's index-out-of-bounds panic; the GPU builder must agree in both cases."""
    w = load_world("weekend-5-80-15"); w.algorithm, w.grid_size = "grid", float(size)
    w.gpu_build = False
    try:
        w.init(); ok = True
    except RtError:
        ok = False
    if ok:
        compare(w, size)
    else:
        w.gpu_build = True; w.init()
        with pytest.raises(RtError):
            w.upload()


def test_gpu_grid_renders_like_host_grid(gpu):
    w = load_world("weekend-5-80-15"); w.algorithm, w.grid_size = "grid", 5.0
    rays_o = np.tile([13.0, 2.0, 3.0], (4096, 1)); rng = np.random.default_rng(3)
    rays_d = rng.normal(size=(4096, 3)) * [1, .3, 1] - [13, 2, 3]
    w.gpu_build = False; w.init(); w.upload(); a = w.hit(rays_o, rays_d)
    w.gpu_build = True; w.init(); w.upload(); b = w.hit(rays_o, rays_d)
    for k in ("prim", "t", "tp", "hit"):
        assert np.array_equal(a[k], b[k]), k


def test_gpu_grid_full_size_mesh_and_time(gpu):
    w = HittableList.from_ply("scenes/dragon_recon/dragon_vrip_res3.ply", subdivide=2)
    w.algorithm, w.grid_size = "grid", 1.0
    t = time.time(); w.gpu_build = False; w.init(); t_host = time.time() - t
    compare(w, 1.0)                                   # first call pays hipCUB kernel loading
    t_dev = compare(w, 1.0)
    print(f"\ngrid build (size 1), 764704 triangles: host {t_host:.2f} s, GPU {t_dev:.3f} s")


@pytest.mark.parametrize("n", [0, 1])
def test_gpu_grid_of_an_empty_and_a_one_object_scene(gpu, n):
    pd = np.dtype([("type", "<u4"), ("material", "<u4"), ("v", "<f8", (9,))])
    md = np.dtype([("type", "<u4"), ("pad", "<u4"), ("albedo", "<f8", (3,)), ("param", "<f8")])
    p = np.zeros(n, pd)
    if n:
        p["v"][0, :4] = (0, 0, -1, 0.5)
    w = HittableList.from_arrays(p, np.zeros((n, 3)), np.zeros(1, md))
    compare(w, 25)
