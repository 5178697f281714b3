#!/usr/bin/env python3
"""Run ON THE GPU BOX: randomized differential test of the BVH walk's round-3 forms against the CPU oracle, beyond the fixed cases of
tests/test_gpu_parity.py: random scenes (spheres / generic triangles / f32-lattice triangles / mixed; 1..1500 objects; clustered centroids
that force multi-primitive leaves; three build modes) traced with coherent fans, scattered and degenerate rays at two tmax values, under
  * the LDS-resident tree kernel forced for every queue (RT_LDS_TREE=2; generic-pair scenes that fit),
  * the line-paired pair table forced onto small trees (RT_PAIR_LINES_MIN=0; lattice scenes), with and without the DPairN records.
Compares hit flag, primitive, rec.t bits, rec.p parameter bits and all counters.  usage: tools/bvh_stress.py [n_scenes=60] [first_seed=500]
Writes gpurun_out/bvh_stress.json."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import raytracer_amd as rt
from raytracer_amd import HittableList
from raytracer_amd.host import _MAT_DT, _PRIM_DT
from tests.test_gpu_parity import _edge_rays
from tests.util import bits, setup_pair

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500
out = {"scenes": 0, "rays": 0, "mismatches": 0, "counter_mismatches": 0, "by_kind": {}, "cases": []}
t0 = time.time()


def make_scene(rng, kind, n):
    prims = np.zeros(n, _PRIM_DT)
    ctr = np.array([rng.uniform(-10, 10), rng.uniform(-10, 10), rng.uniform(-25, -6)])
    spread = float(rng.choice([0.5, 2.0, 6.0]))
    c = rng.uniform(-spread, spread, size=(n, 3)) + ctr
    tri = c[:, None, :] + rng.normal(scale=rng.choice([1e-6, 0.2, 0.8, 2.5], size=(n, 1, 1)), size=(n, 3, 3))
    flat = rng.random(n) < 0.04
    tri[flat, 2] = tri[flat, 1]
    if kind == "lattice":
        tri = tri.astype(np.float32).astype(np.float64)          # every coordinate an f32: the scale-1 lattice
    is_tri = np.ones(n, bool) if kind in ("triangles", "lattice") else (np.zeros(n, bool) if kind == "spheres" else rng.random(n) < 0.5)
    prims["type"] = is_tri.astype(np.uint32)
    v = tri.reshape(n, 9).copy()
    sph = ~is_tri
    v[sph, :3] = c[sph]; v[sph, 3] = rng.choice([0.05, 0.3, 1.0, 3.0], size=int(sph.sum())) * rng.uniform(0.5, 1.5, size=int(sph.sum())); v[sph, 4:] = 0.0
    prims["v"] = v
    mats = np.zeros(3, _MAT_DT); mats["type"] = [0, 1, 2]; mats["albedo"] = rng.uniform(0.2, 1.0, size=(3, 3)); mats["param"] = [0, 0.4, 1.5]
    prims["material"] = rng.integers(0, 3, size=n)
    cent = np.where(is_tri[:, None], ((tri[:, 0] + tri[:, 1]) + tri[:, 2]) / 3.0, c)
    if rng.random() < 0.35:                                      # clustered centroids: the SAH cannot separate them -> leaves of several primitives
        cent = np.round(cent / 1.5) * 1.5
    return HittableList.from_arrays(prims, cent, mats, filename="scenes/stress.json"), ctr


for s in range(seed0, seed0 + n_scenes):
    rng = np.random.default_rng(s)
    kind = str(rng.choice(["spheres", "triangles", "lattice", "lattice", "mixed"]))
    n = int(rng.choice([1, 2, 3, 5, 17, 64, 200, 487, 600, 1500]))
    mode = str(rng.choice(["sah-position", "sah-position", "sah-plane", "naive"]))
    w, ctr = make_scene(rng, kind, n)
    osc = setup_pair(w, "bvh", mode)
    o1 = np.tile(ctr + np.array([[0.0, 0.0, 12.0]]), (3000, 1)); d1 = np.array([0.0, 0.0, -1.0]) + rng.normal(scale=0.4, size=(3000, 3))
    o2 = rng.uniform(-9, 9, size=(5000, 3)) + ctr; d2 = rng.normal(size=(5000, 3))
    o3, d3 = _edge_rays(w, 1000, s + 3)
    o = np.concatenate([o1, o2, o3]); d = np.concatenate([d1, d2, d3])
    with np.errstate(all="ignore"):
        refs = {tmax: osc.trace("bvh", o, d, 0.001, tmax) for tmax in (float("inf"), 5.0)}
    variants = [("lds", {"RT_LDS_TREE": "2"}), ("plain", {"RT_LDS_TREE": "0", "RT_PAIR_LINES": "0"})]
    if kind == "lattice":
        variants = [("paired", {"RT_PAIR_LINES_MIN": "0"}), ("paired+pairn", {"RT_PAIR_LINES_MIN": "0", "RT_PAIRN": "1"}), ("plain", {"RT_PAIR_LINES": "0"})]
    bad = cbad = 0; layouts = []
    for label, env in variants:
        for k, v in env.items():
            os.environ[k] = v
        rt.init(0); w.upload(); layouts.append(f"{label}:{w.bvh_layout()}")
        with np.errstate(all="ignore"):
            for tmax, r in refs.items():
                g = w.hit(o, d, 0.001, tmax)
                m = (g["hit"] != r["hit"]) | (g["prim"] != r["prim"]) | (bits(g["t"]) != bits(r["t"])) | (bits(g["tp"]) != bits(r["tp"]))
                bad += int(m.sum())
                for k in ("intersection_checks", "traversal_steps", "overlapping_aabb", "box_tests", "prim_tests"):
                    cbad += int(g["stats"][k] != r["stats"][k])
                out["rays"] += len(o)
        for k in env:
            del os.environ[k]
    rt.init(0)
    nd = w.bvh.nodes()
    max_leaf = int(nd["count"][nd["is_leaf"].astype(bool)].max())
    out["scenes"] += 1; out["mismatches"] += bad; out["counter_mismatches"] += cbad
    out["by_kind"][kind] = out["by_kind"].get(kind, 0) + 1
    out["cases"].append({"seed": s, "kind": kind, "objects": n, "mode": mode, "nodes": int(len(nd)), "depth": int(w.bvh.depth), "max_leaf": max_leaf, "layouts": layouts,
                         "mismatches": bad, "counter_mismatches": cbad})
    print(f"seed {s}: {kind} n={n} {mode} nodes {len(nd)} depth {w.bvh.depth} max leaf {max_leaf} {layouts} mismatches {bad} counters {cbad}", flush=True)
    w.close()
out["seconds"] = round(time.time() - t0, 1)
os.makedirs("gpurun_out", exist_ok=True); json.dump(out, open("gpurun_out/bvh_stress.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "cases"}))
sys.exit(1 if out["mismatches"] or out["counter_mismatches"] else 0)
