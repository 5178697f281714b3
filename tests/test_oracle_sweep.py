"""The oracle's two statements of the SAH-position build.

`build_bvh()` is the literal restatement of bvh.rs:86-160 (every candidate split re-folds both sides: O(n^2) per node);
`build_bvh(sweep=True)` states the same definitions with prefix boxes.  Here: byte-identical node arrays and object orders on
every scene the literal form finishes, and then the sweep form — an implementation that shares no code with the product's
builder (raytracer_amd/csrc/host/accel.cpp) — pins the product's host-built trees for the meshes above that size: res3
(47 794 triangles) and the bench's 764 704-triangle mesh.  The GPU builder is held to the host builder in
tests/test_gpu_build.py, and to this oracle in the same file."""
import numpy as np
import pytest

from raytracer_amd import HittableList

from .util import SCENES, load_world, oracle_of, weekend_mixed

SMALL = [s for s in SCENES if s != "dragon_res3"]


@pytest.mark.parametrize("scene", SMALL + ["mixed"])
def test_sweep_equals_literal_builder(scene):
    w = weekend_mixed() if scene == "mixed" else load_world(scene)
    a = oracle_of(w).build_bvh("sah-position")
    b = oracle_of(w).build_bvh("sah-position", sweep=True)
    na, oa = a.get_bvh(); nb, ob = b.get_bvh()
    assert len(na) == len(nb) and na.tobytes() == nb.tobytes(), f"{scene}: node arrays differ"
    assert np.array_equal(oa, ob), f"{scene}: object order differs"


def test_sweep_equals_literal_on_adversarial_inputs():
    """Duplicate centroids (sort stability is observable), coincident boxes, zero-area boxes, signed zeros."""
    rng = np.random.default_rng(5)
    from oracle.oracle_py import MAT_DT, PRIM_DT, OracleScene
    n = 600
    prims = np.zeros(n, PRIM_DT); prims["type"] = 1
    grid = rng.integers(-3, 4, size=(n, 3)).astype(np.float64)          # many equal coordinates
    v = grid[:, None, :] + rng.choice([0.0, -0.0, 0.5, 1.0], size=(n, 3, 3))
    v[::7] = np.where(v[::7] == 0.0, -0.0, v[::7])
    prims["v"] = v.reshape(n, 9)
    cent = np.round(v.mean(axis=1))                                     # ties everywhere
    mats = np.zeros(1, MAT_DT)
    a = OracleScene(prims, cent, mats).build_bvh("sah-position")
    b = OracleScene(prims, cent, mats).build_bvh("sah-position", sweep=True)
    na, oa = a.get_bvh(); nb, ob = b.get_bvh()
    assert na.tobytes() == nb.tobytes() and np.array_equal(oa, ob)


@pytest.mark.parametrize("scene,subdivide", [("dragon_res3", 0), ("dragon_res3", 2)])
def test_host_builder_pinned_by_sweep_oracle_at_scale(scene, subdivide):
    """north_star: "the flattened BVH node layout must be bit-exact against the reference's SAH build" — at the sizes the
    bench runs (subdivide=2 is bench.py's 764 704-triangle mesh)."""
    w = HittableList.from_ply(SCENES[scene][0], subdivide=subdivide)
    w.algorithm, w.bvh_mode = "bvh", "sah-position"
    w.init()
    o = oracle_of(w).build_bvh("sah-position", sweep=True)
    nodes, order = o.get_bvh()
    assert len(nodes) == w.bvh.n_nodes
    assert nodes.tobytes() == w.bvh.nodes().tobytes(), "node array differs from the oracle's"
    assert np.array_equal(order, w.bvh.order(len(w))), "object order differs from the oracle's"
