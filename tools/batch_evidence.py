#!/usr/bin/env python3
"""Run ON THE GPU BOX: a 3840x2160 render whose samples do not fit one batch (256 spp = two 128-spp batches at 75 % of HBM) against
the same samples rendered as two calls: the multi-batch path at the bench's size."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import raytracer_amd as rt
from raytracer_amd import Camera, HittableList
rt.init(0); rt.set_profiling(True)
fn = "scenes/dragon_recon/dragon_vrip_res3.ply"
w = HittableList.from_ply(fn, subdivide=2); w.algorithm = "bvh"; w.init(); w.upload()
cam = Camera(fn, image_width=3840, samples_per_pixel=256)
t = time.time(); whole, st = cam.render(w, seed=1); t_whole = time.time() - t
a, sa = cam.render(w, seed=1, s_begin=0, s_count=128); b, sb = cam.render(w, seed=1, s_begin=128, s_count=128)
rel = np.abs(whole - (a + b)) / np.maximum(np.abs(whole), 1e-300)
r = {"image": f"{cam.image_width}x{cam.image_height}", "spp": 256, "extend_launches": int(st["extend_launches"]), "seconds": t_whole,
     "Mrays_per_s": (st["primary_rays"] + st["scatter_rays"]) / (st["render_ms"] / 1e3) / 1e6,
     "max_rel_diff_vs_two_calls": float(rel.max()), "scatter_rays_equal": int(st["scatter_rays"]) == int(sa["scatter_rays"]) + int(sb["scatter_rays"]),
     "intersection_checks_equal": int(st["intersection_checks"]) == int(sa["intersection_checks"]) + int(sb["intersection_checks"])}
print(json.dumps(r)); json.dump(r, open("gpurun_out/batch_evidence.json", "w"), indent=1)
