#!/usr/bin/env python3
"""Per-bounce kernel times (RT_BOUNCE_LOG=1) of one render: tools/bounce_log.py <scene> <algorithm> <grid_size> <width> <spp> [max_depth]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
os.environ["RT_BOUNCE_LOG"] = "1"
import raytracer_amd as rt
from raytracer_amd import Camera, HittableList
scene, algo, gs, width, spp = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
md = int(sys.argv[6]) if len(sys.argv) > 6 else 50
rt.init(0); rt.set_profiling(True)
w = HittableList.from_ply(scene) if scene.endswith(".ply") else HittableList.from_json(scene)
w.algorithm, w.grid_size = algo, gs
w.init().upload()
cam = Camera(scene, image_width=width, samples_per_pixel=spp, max_depth=md)
cam.render(w, seed=1)          # warm-up (allocations)
print(f"--- {scene} {algo} {gs} {width} {spp}", file=sys.stderr, flush=True)
_, st = cam.render(w, seed=1)
print({k: st[k] for k in ("render_ms", "extend_ms", "shade_ms", "generate_ms", "ray_segments", "prim_tests", "box_tests", "traversal_steps", "scatter_rays")}, file=sys.stderr)
