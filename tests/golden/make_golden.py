#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (run from the repo root: python tests/golden/make_golden.py).

The reference holds no output fixtures for this path and cannot be built here (Rust), so these vectors pin the
ORACLE's behaviour (a regression net for later edits), not the reference's: parity stays "unpinned" upstream.
Scene inputs are parsed here with Python's own json / numpy float32 readers, independently of the product's C++
loaders, so the fixtures also check those loaders."""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from oracle import oracle_py as O  # noqa: E402

MAT_TYPES = {"Lambertian": 0, "Metal": 1, "Dielectric": 2}


def parse_json_scene(path):
    """{"objects":[{"type":"Sphere"|"Triangle",...,"mat":{...}}]} (reference: main.rs:36-41, sphere.rs:12-17, triangle.rs:11-18)"""
    objs = json.load(open(path))["objects"]
    prims = np.zeros(len(objs), O.PRIM_DT); cen = np.zeros((len(objs), 3)); mats = np.zeros(len(objs), O.MAT_DT)
    for i, o in enumerate(objs):
        m = o["mat"]
        mats[i]["type"] = MAT_TYPES[m["type"]]
        if "albedo" in m: mats[i]["albedo"] = m["albedo"]["e"]
        mats[i]["param"] = m.get("fuzz", m.get("refraction_index", 0.0))
        prims[i]["material"] = i
        if o["type"] == "Sphere":
            prims[i]["type"] = 0; prims[i]["v"][:3] = o["center"]["e"]; prims[i]["v"][3] = o["radius"]; cen[i] = o["center"]["e"]
        else:
            prims[i]["type"] = 1; prims[i]["v"] = o["v0"]["e"] + o["v1"]["e"] + o["v2"]["e"]; cen[i] = o["centroid"]["e"]
    return prims, cen, mats


def parse_ply_scene(path):
    """ASCII PLY -> triangles: f32 vertices widened and scaled by 100 (utils/parser.rs:67-73), Lambertian(0.8,0.2,0.1) (main.rs:43)."""
    lines = open(path).read().split("\n")
    nv = nf = 0; i = 0
    while lines[i].strip() != "end_header":
        t = lines[i].split()
        if t[:2] == ["element", "vertex"]: nv = int(t[2])
        if t[:2] == ["element", "face"]: nf = int(t[2])
        i += 1
    i += 1
    v = np.array([l.split()[:3] for l in lines[i:i + nv]], dtype=np.float32).astype(np.float64) * 100.0
    f = np.array([l.split()[1:4] for l in lines[i + nv:i + nv + nf]], dtype=np.int64)
    prims = np.zeros(nf, O.PRIM_DT); prims["type"] = 1
    prims["v"] = np.concatenate([v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]], axis=1)
    cen = np.stack([(v[f[:, 0], c] + v[f[:, 1], c] + v[f[:, 2], c]) / 3.0 for c in range(3)], axis=1)   # triangle.rs:22-26
    mats = np.zeros(1, O.MAT_DT); mats[0]["albedo"] = (0.8, 0.2, 0.1)
    return prims, cen, mats


CAMERAS = {   # main.rs:72-104
    "default": dict(vfov=90.0, look_from=(0, 0, 0), look_at=(0, 0, -1), focus_dist=1.0),
    "weekend": dict(vfov=20.0, look_from=(13, 2, 3), look_at=(0, 0, 0), focus_dist=1.0),
    "dragon": dict(vfov=20.0, look_from=(0, 15, 50), look_at=(0, 12, 0), focus_dist=50.0),
}


def camera(kind, width):
    c = CAMERAS[kind]
    return O.camera(16.0 / 9.0, width, c["vfov"], c["look_from"], c["look_at"], (0, 1, 0), 0.1, c["focus_dist"])


CASES = [  # name, path, loader, camera, [(algorithm, bvh_mode, grid_size)]
    ("simple_fuzzy_metal", "scenes/simple_fuzzy_metal.json", parse_json_scene, "default",
     [("naive", None, None), ("bvh", "sah-position", None), ("bvh", "sah-plane", None), ("bvh", "naive", None), ("grid", None, 25.0), ("grid", None, 5.0)]),
    ("simple_hollow_glass", "scenes/simple_hollow_glass.json", parse_json_scene, "default", [("bvh", "sah-position", None)]),
    ("triangle_test", "scenes/triangle_test.json", parse_json_scene, "default", [("naive", None, None), ("bvh", "sah-position", None)]),
    ("triangle_materials", "scenes/triangle_materials.json", parse_json_scene, "default",
     [("naive", None, None), ("bvh", "sah-position", None), ("grid", None, 5.0)]),
    ("weekend-final", "scenes/weekend-final.json", parse_json_scene, "weekend", [("bvh", "sah-position", None), ("grid", None, 25.0)]),
    ("dragon_res4", "scenes/dragon_recon/dragon_vrip_res4.ply", parse_ply_scene, "dragon", [("bvh", "sah-plane", None), ("grid", None, 5.0)]),
]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = {}
    meta = {}
    for name, path, loader, camkind, algos in CASES:
        prims, cen, mats = loader(path)
        meta[name] = {"scene_sha256": sha(prims) + sha(cen) + sha(mats), "n": int(len(prims))}
        cam = camera(camkind, 64)
        jj, ii = np.meshgrid(np.arange(cam.image_height), np.arange(cam.image_width), indexing="ij")
        o, d = O.generate(cam, 1, 0, ii.ravel(), jj.ravel())
        out[f"{name}/rays_o"] = o[:8]; out[f"{name}/rays_d"] = d[:8]
        sc = O.OracleScene(prims, cen, mats)
        for algo, mode, gs in algos:
            key = f"{name}/{algo}" + (f"-{mode}" if mode else "") + (f"-{gs:g}" if gs else "")
            if algo == "bvh":
                sc.build_bvh(mode)
                nodes, order = sc.get_bvh()
                meta[key] = {"bvh_nodes": int(len(nodes)), "bvh_sha256": sha(nodes) + sha(order)}
                if len(nodes) <= 16:
                    out[key + "/nodes"] = nodes; out[key + "/order"] = order
            if algo == "grid":
                sc.build_grid(gs)
                aabb, off, pl = sc.get_grid()
                meta[key] = {"grid_sha256": sha(aabb) + sha(off) + sha(pl), "list_len": int(len(pl))}
            r = sc.trace(algo, o, d)
            out[key + "/prim"] = r["prim"].astype(np.int32); out[key + "/t"] = r["t"]; out[key + "/hit"] = r["hit"]
            meta.setdefault(key, {})["stats"] = r["stats"]
            cam2 = camera(camkind, 32)
            img, cs = sc.render(cam2, algo, 0, 2, 50, seed=1, nthreads=8)
            out[key + "/img"] = img; meta[key]["render_stats"] = cs
    np.savez_compressed("tests/golden/oracle_golden.npz", **out)
    json.dump(meta, open("tests/golden/oracle_golden.json", "w"), indent=1, sort_keys=True)
    print("wrote", len(out), "arrays;", os.path.getsize("tests/golden/oracle_golden.npz"), "bytes")


if __name__ == "__main__":
    main()
