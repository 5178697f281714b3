#!/usr/bin/env python3
"""Run ON THE GPU BOX: same-seed full path tracing, GPU vs CPU oracle, at 480x270: per-pixel radiance sums (relative error) and
all Data counters (must be equal: every hit decision of every bounce is the oracle's).  Writes gpurun_out/same_seed_evidence.json."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import raytracer_amd as rt
from raytracer_amd import Camera, HittableList
from oracle import oracle_py as O
rt.init(0)
out = []
# default: the three BVH scenes; or arguments "scene:algorithm:grid_size:spp", e.g. dragon_res4:grid:5:16
jobs = [tuple(a.split(":")) for a in sys.argv[1:]] or [("weekend-final", "bvh", "0", "64"), ("weekend-mixed", "bvh", "0", "64"), ("dragon", "bvh", "0", "32")]
for name, algo, gsize, spp in jobs:
    spp = int(spp)
    if name == "dragon_res4":
        fn = "scenes/dragon_recon/dragon_vrip_res4.ply"; w = HittableList.from_ply(fn)
    elif name == "dragon":
        fn = "scenes/dragon_recon/dragon_vrip_res3.ply"; w = HittableList.from_ply(fn, subdivide=2)
    elif name == "weekend-mixed":
        fn = "scenes/weekend-mixed.json"; w = HittableList.from_json("scenes/weekend-final.json"); w.append(HittableList.from_json("scenes/triangle_materials.json"))
    else:
        fn = f"scenes/{name}.json"; w = HittableList.from_json(fn)
    w.algorithm = algo; w.grid_size = float(gsize) if algo == "grid" else 25.0; w.init(); w.upload()
    cam = Camera(fn, image_width=480, samples_per_pixel=spp)
    img, st = cam.render(w, seed=7)
    osc = O.OracleScene(w.prims(), w.centroids(), w.materials())
    if algo == "bvh": osc.build_bvh("sah-position", sweep=True)
    if algo == "grid": osc.build_grid(float(gsize))
    oc = O.camera_from_rt(cam.c); H = cam.image_height
    ref = np.zeros_like(img); cnt = {}; t = time.time()
    for r0 in range(0, H, 30):
        band, c = osc.render(oc, algo, 0, spp, cam.max_depth, seed=7, rows=(r0, min(H, r0 + 30)), nthreads=16)
        ref += band
        for k, v in c.items(): cnt[k] = cnt.get(k, 0) + int(v)
        print(f"{name}: CPU rows {min(H, r0 + 30)}/{H}, {time.time() - t:.0f} s", flush=True)
    keys = ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb", "gridbox_intersection_checks")
    rel = np.abs(img - ref) / np.maximum(np.abs(ref), 1e-300)
    r = {"scene": name, "algorithm": algo + (f" size={gsize}" if algo == "grid" else ""), "image": f"480x{H}", "spp": spp, "seed": 7, "max_rel_error_of_pixel_sums": float(rel[ref != 0].max()) if (ref != 0).any() else 0.0,
         "pixels_bit_identical": int((img == ref).all(axis=2).sum()), "pixels": int(img.shape[0] * img.shape[1]),
         "counters_gpu": {k: int(st[k]) for k in keys}, "counters_oracle": {k: cnt.get(k, 0) for k in keys}}
    r["counters_equal"] = r["counters_gpu"] == r["counters_oracle"]
    out.append(r); print(json.dumps(r), flush=True)
    w.close()
os.makedirs("gpurun_out", exist_ok=True); json.dump(out, open("gpurun_out/same_seed_evidence" + ("_args" if sys.argv[1:] else "") + ".json", "w"), indent=1)
