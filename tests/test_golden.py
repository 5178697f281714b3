"""Committed golden vectors (tests/golden/, made by make_golden.py with the oracle) against
 (a) the oracle fed by the PRODUCT's C++ scene loaders — CPU, and
 (b) the HIP path through the C ABI — GPU.
The fixtures were produced from scenes parsed by Python's json / numpy-float32 readers, so (a) also proves the C++
JSON and PLY loaders deliver bit-identical scene arrays."""
import hashlib
import json

import numpy as np
import pytest

from oracle import oracle_py as O
from raytracer_amd import Camera

from .util import bits, load_world, oracle_of

G = np.load("tests/golden/oracle_golden.npz")
META = json.load(open("tests/golden/oracle_golden.json"))
CAM_WIDTH_TRACE, CAM_WIDTH_IMG = 64, 32


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def keys():
    return sorted({k.rsplit("/", 1)[0] for k in G.files if k.endswith("/prim")})


def split_key(key):
    scene, rest = key.split("/")
    parts = rest.split("-")
    algo = parts[0]
    mode = "-".join(parts[1:]) if algo == "bvh" else None
    gs = float(parts[1]) if algo == "grid" else None
    return scene, algo, mode, gs


def primary(cam_c):
    jj, ii = np.meshgrid(np.arange(cam_c.image_height), np.arange(cam_c.image_width), indexing="ij")
    return ii.ravel().astype(np.uint32), jj.ravel().astype(np.uint32)


@pytest.mark.parametrize("scene", sorted({k.split("/")[0] for k in keys()}))
def test_cpp_loaders_match_python_parse(scene):
    w = load_world(scene)
    assert len(w) == META[scene]["n"]
    assert sha(w.prims()) + sha(w.centroids()) + sha(w.materials()) == META[scene]["scene_sha256"]


@pytest.mark.parametrize("key", keys())
def test_oracle_reproduces_golden(key):
    scene, algo, mode, gs = split_key(key)
    w = load_world(scene)
    osc = oracle_of(w)
    cam = Camera(w.filename, image_width=CAM_WIDTH_TRACE)
    oc = O.camera_from_rt(cam.c)
    pi, pj = primary(oc)
    o, d = O.generate(oc, 1, 0, pi, pj)
    assert np.array_equal(bits(o[:8]), bits(G[f"{scene}/rays_o"])) and np.array_equal(bits(d[:8]), bits(G[f"{scene}/rays_d"]))
    if algo == "bvh":
        osc.build_bvh(mode)
        nodes, order = osc.get_bvh()
        assert sha(nodes) + sha(order) == META[key]["bvh_sha256"]
        if key + "/nodes" in G.files:
            assert nodes.tobytes() == G[key + "/nodes"].tobytes() and np.array_equal(order, G[key + "/order"])
    if algo == "grid":
        osc.build_grid(gs)
        a, off, pl = osc.get_grid()
        assert sha(a) + sha(off) + sha(pl) == META[key]["grid_sha256"]
    r = osc.trace(algo, o, d)
    assert np.array_equal(r["prim"], G[key + "/prim"]) and np.array_equal(bits(r["t"]), bits(G[key + "/t"]))
    assert np.array_equal(r["hit"], G[key + "/hit"]) and r["stats"] == META[key]["stats"]
    cam2 = O.camera_from_rt(Camera(w.filename, image_width=CAM_WIDTH_IMG).c)
    img, cs = osc.render(cam2, algo, 0, 2, 50, seed=1, nthreads=4)
    assert np.array_equal(bits(img), bits(G[key + "/img"])) and cs == META[key]["render_stats"]


@pytest.mark.gpu
@pytest.mark.parametrize("key", keys())
def test_hip_path_reproduces_golden(gpu, key):
    scene, algo, mode, gs = split_key(key)
    w = load_world(scene)
    w.algorithm = algo
    if mode: w.bvh_mode = mode
    if gs: w.grid_size = gs
    w.init().upload()
    if algo == "bvh":
        assert sha(w.bvh.nodes()) + sha(w.bvh.order(len(w))) == META[key]["bvh_sha256"]
    if algo == "grid":
        assert sha(w.grid.cell_aabb()) + sha(w.grid.cell_offset()) + sha(w.grid.prim_list()) == META[key]["grid_sha256"]
    cam = Camera(w.filename, image_width=CAM_WIDTH_TRACE)
    pi, pj = primary(cam.c)
    o, d = cam.get_rays(pi, pj, sample=0, seed=1)
    assert np.array_equal(bits(o[:8]), bits(G[f"{scene}/rays_o"])) and np.array_equal(bits(d[:8]), bits(G[f"{scene}/rays_d"]))
    r = w.hit(o, d)
    assert np.array_equal(r["prim"], G[key + "/prim"]) and np.array_equal(bits(r["t"]), bits(G[key + "/t"]))
    assert np.array_equal(r["hit"], G[key + "/hit"])
    for k, v in META[key]["stats"].items():
        if k in r["stats"] and k not in ("primary_rays", "scatter_rays"):
            assert r["stats"][k] == v, k
    cam2 = Camera(w.filename, image_width=CAM_WIDTH_IMG, samples_per_pixel=2)
    img, st = cam2.render(w, seed=1)
    ref = G[key + "/img"]
    assert np.max(np.abs(img - ref) / np.maximum(np.abs(ref), 1e-300)) <= 1e-12
    for k, v in META[key]["render_stats"].items():
        if k in st and k not in ("box_tests", "prim_tests"):
            assert st[k] == v, k
    w.close()
