"""Bvh::new on the GPU (rt_bvh_build / rt_bvh_build_mode) against the host builder and against the oracle (its literal
restatement of bvh.rs:19-180 up to res4, its sweep form above that — tests/test_oracle_sweep.py): nodes, numbering and object
order byte for byte."""
import time

import numpy as np
import pytest

from raytracer_amd import HittableList

from .util import load_world, oracle_of, weekend_mixed

pytestmark = pytest.mark.gpu


def both(w, mode="sah-position"):
    w.algorithm, w.bvh_mode = "bvh", mode
    w.gpu_build = False; w.init(); host = (w.bvh.nodes().copy(), w.bvh.order(len(w)).copy(), w.bvh.depth)
    w.gpu_build = True; t = time.time(); w.init(); dt = time.time() - t
    return host, (w.bvh.nodes().copy(), w.bvh.order(len(w)).copy(), w.bvh.depth), dt


@pytest.mark.parametrize("scene", ["simple_fuzzy_metal", "triangle_test", "triangle_materials", "weekend-5-80-15", "weekend-final", "mixed",
                                   "dragon_res4", "dragon_res3"])
def test_gpu_bvh_equals_host_bvh(gpu, scene):
    w = weekend_mixed() if scene == "mixed" else load_world(scene)
    host, dev, _ = both(w)
    assert dev[2] == host[2], "depth"
    assert len(dev[0]) == len(host[0]), "node count"
    assert np.array_equal(dev[1], host[1]), f"object order differs at {np.flatnonzero(dev[1] != host[1])[:8]}"
    for f in ("is_leaf", "left", "right", "first", "count"):
        assert np.array_equal(dev[0][f], host[0][f]), f
    assert dev[0].tobytes() == host[0].tobytes(), "box planes differ"
    on, oo = oracle_of(w).build_bvh("sah-position", sweep=scene == "dragon_res3").get_bvh()
    assert on.tobytes() == dev[0].tobytes() and np.array_equal(oo, dev[1]), "GPU-built tree differs from the oracle's"


@pytest.mark.parametrize("mode", ["sah-plane", "naive"])
@pytest.mark.parametrize("scene", ["simple_fuzzy_metal", "triangle_test", "triangle_materials", "weekend-5-80-15", "weekend-final", "mixed",
                                   "dragon_res4", "dragon_res3"])
def test_gpu_bvh_other_modes_equal_host_and_oracle(gpu, scene, mode):
    """rt_bvh_build_mode for `-o bvh-sah-plane` (bvh.rs:114-123) and `-o bvh-naive` (bvh.rs:87-99): nodes, numbering and object
    order byte for byte as the host builder's and the oracle's literal builder's (which stays cheap in these modes: one
    candidate per axis / none)."""
    w = weekend_mixed() if scene == "mixed" else load_world(scene)
    host, dev, _ = both(w, mode)
    assert dev[2] == host[2] and len(dev[0]) == len(host[0])
    assert np.array_equal(dev[1], host[1]), f"object order differs at {np.flatnonzero(dev[1] != host[1])[:8]}"
    assert dev[0].tobytes() == host[0].tobytes()
    on, oo = oracle_of(w).build_bvh(mode).get_bvh()
    assert on.tobytes() == dev[0].tobytes() and np.array_equal(oo, dev[1]), "GPU-built tree differs from the oracle's"


@pytest.mark.parametrize("mode", ["sah-plane", "naive"])
def test_gpu_bvh_other_modes_small_and_adversarial(gpu, mode):
    """Node sizes around the modes' thresholds (naive leaves nodes of < 3 objects alone, order included), duplicate centroids,
    signed zeros."""
    from oracle.oracle_py import OracleScene
    dt = [("type", "<u4"), ("material", "<u4"), ("v", "<f8", (9,))]
    mats = np.zeros(1, dtype=[("type", "<u4"), ("pad", "<u4"), ("albedo", "<f8", (3,)), ("param", "<f8")])
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 33, 600):
        p = np.zeros(n, dtype=dt); p["type"] = 1
        grid = rng.integers(-3, 4, size=(n, 3)).astype(np.float64)
        v = grid[:, None, :] + rng.choice([0.0, -0.0, 0.5, 1.0], size=(n, 3, 3))
        v[::3] = np.where(v[::3] == 0.0, -0.0, v[::3])
        p["v"] = v.reshape(n, 9)
        cent = np.round(v.mean(axis=1))[::-1].copy()         # descending-ish x: the sort has work to do, ties abound
        w = HittableList.from_arrays(p, cent, mats)
        host, dev, _ = both(w, mode)
        on, oo = OracleScene(w.prims(), w.centroids(), w.materials()).build_bvh(mode).get_bvh()
        assert np.array_equal(dev[1], oo) and np.array_equal(host[1], oo), f"n={n}: object order"
        assert host[0].tobytes() == on.tobytes(), f"n={n}: host vs oracle"
        assert dev[0].tobytes() == on.tobytes(), f"n={n}: GPU vs oracle"


def test_gpu_bvh_single_object_and_duplicates(gpu):
    one = np.zeros(1, dtype=[("type", "<u4"), ("material", "<u4"), ("v", "<f8", (9,))]); one["v"][0, :4] = (0, 0, -1, 0.5)
    mats = np.zeros(1, dtype=[("type", "<u4"), ("pad", "<u4"), ("albedo", "<f8", (3,)), ("param", "<f8")])
    w = HittableList.from_arrays(one, np.array([[0.0, 0.0, -1.0]]), mats)
    host, dev, _ = both(w)
    assert dev[0].tobytes() == host[0].tobytes() and len(dev[0]) == 1
    # many objects sharing centroid coordinates: stable-sort ties decide the layout
    rng = np.random.default_rng(5); n = 2000
    p = np.zeros(n, dtype=one.dtype); c = rng.integers(-3, 4, size=(n, 3)).astype(np.float64)
    p["v"][:, :3] = c; p["v"][:, 3] = rng.uniform(0.1, 0.6, size=n)
    w = HittableList.from_arrays(p, c, mats)
    host, dev, _ = both(w)
    assert np.array_equal(dev[1], host[1]) and dev[0].tobytes() == host[0].tobytes()


def test_gpu_bvh_signed_zeros_and_coincident_boxes(gpu):
    """The adversarial input of tests/test_oracle_sweep.py (duplicate centroids, coincident and zero-area boxes, +0 / -0 planes)
    through rt_bvh_build: the right child's box is a left fold of its slice (bvh.rs:61-64,150-151), and of two equal planes
    f64::min / f64::max keep the later object's — the sign of a zero plane included.  Held to the oracle's LITERAL builder."""
    from oracle.oracle_py import OracleScene
    rng = np.random.default_rng(5); n = 600
    p = np.zeros(n, dtype=[("type", "<u4"), ("material", "<u4"), ("v", "<f8", (9,))]); p["type"] = 1
    grid = rng.integers(-3, 4, size=(n, 3)).astype(np.float64)
    v = grid[:, None, :] + rng.choice([0.0, -0.0, 0.5, 1.0], size=(n, 3, 3))
    v[::7] = np.where(v[::7] == 0.0, -0.0, v[::7])
    p["v"] = v.reshape(n, 9)
    cent = np.round(v.mean(axis=1))
    mats = np.zeros(1, dtype=[("type", "<u4"), ("pad", "<u4"), ("albedo", "<f8", (3,)), ("param", "<f8")])
    assert np.signbit(v[v == 0.0]).any() and not np.signbit(v[v == 0.0]).all(), "the input must hold both zeros"
    w = HittableList.from_arrays(p, cent, mats)
    host, dev, _ = both(w)
    on, oo = OracleScene(w.prims(), w.centroids(), w.materials()).build_bvh("sah-position").get_bvh()
    assert np.array_equal(dev[1], oo), "object order"
    assert host[0].tobytes() == on.tobytes(), "host builder vs the oracle's literal builder"
    bad = [i for i in range(len(on)) if dev[0][i].tobytes() != on[i].tobytes()]
    assert not bad, f"GPU-built nodes {bad[:8]} differ from the oracle's literal builder: {dev[0][bad[0]]} vs {on[bad[0]]}"


def test_gpu_bvh_full_size_mesh_and_time(gpu):
    """The bench mesh (764 704 triangles): identical to the host build; the time is printed for DESIGN.md."""
    w = HittableList.from_ply("scenes/dragon_recon/dragon_vrip_res3.ply", subdivide=2)
    w.algorithm = "bvh"
    t = time.time(); w.gpu_build = False; w.init(); t_host = time.time() - t
    host = (w.bvh.nodes().tobytes(), w.bvh.order(len(w)).tobytes(), w.bvh.depth)
    w.gpu_build = True; w.init()                         # first call pays hipCUB kernel loading
    t = time.time(); w.init(); t_dev = time.time() - t
    assert (w.bvh.nodes().tobytes(), w.bvh.order(len(w)).tobytes(), w.bvh.depth) == host
    on, oo = oracle_of(w).build_bvh("sah-position", sweep=True).get_bvh()
    assert on.tobytes() == host[0] and oo.tobytes() == host[1], "the bench mesh's tree differs from the oracle's sweep build"
    print(f"\nBVH build, 764704 triangles: host {t_host:.2f} s, GPU {t_dev:.3f} s")
