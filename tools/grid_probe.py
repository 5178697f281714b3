#!/usr/bin/env python3
"""One grid render of dragon res4 for profiling: python3 tools/grid_probe.py <grid size> [spp]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import raytracer_amd as rt
from raytracer_amd import Camera, HittableList
size = float(sys.argv[1]) if len(sys.argv) > 1 else 25.0
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rt.init(0); rt.set_profiling(True)
w = HittableList.from_ply("scenes/dragon_recon/dragon_vrip_res4.ply")
w.algorithm, w.grid_size = "grid", size
w.init(); w.upload()
cam = Camera(w.filename, image_width=900, samples_per_pixel=spp)
_, st = cam.render(w, seed=1)
print({k: st[k] for k in ("primary_rays", "scatter_rays", "prim_tests", "traversal_steps", "extend_ms", "render_ms")})
