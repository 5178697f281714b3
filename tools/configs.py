#!/usr/bin/env python3
"""Run BASELINE.json's configs[1], [2] and [4] on one GPU and record Mrays/s (configs[3] is bench.py; configs[0] is the CPU
parity case).  Usage (GPU box): python tools/configs.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
import raytracer_amd as rt
from raytracer_amd import Camera, HittableList

rt.init(0); rt.set_profiling(True)
out = []

def run(label, world, cam_name, algo, width, spp, grid_size=None, repeat=2):
    world.algorithm = algo
    if grid_size: world.grid_size = grid_size
    t = time.time(); world.init(); world.upload(); init_s = time.time() - t
    cam = Camera(cam_name, image_width=width, samples_per_pixel=spp)
    best = None
    for _ in range(repeat):
        _, st = cam.render(world, seed=1)
        if best is None or st["render_ms"] < best["render_ms"]: best = st
    rays = best["primary_rays"] + best["scatter_rays"]
    r = {"config": label, "algorithm": algo + (f" size={grid_size:g}" if grid_size else ""), "image": f"{cam.image_width}x{cam.image_height}", "spp": spp,
         "primitives": len(world), "init_s": round(init_s, 3), "render_ms": best["render_ms"], "Mrays_per_s": rays / best["render_ms"] / 1e3,
         "rays": rays, "extend_ms": best["extend_ms"], "shade_ms": best["shade_ms"], "generate_ms": best["generate_ms"],
         "intersection_checks_per_ray": best["intersection_checks"] / max(best["ray_segments"], 1)}
    out.append(r); print(json.dumps(r), flush=True)

w = HittableList.from_json("scenes/weekend-final.json")
run("configs[1] weekend-final BVH 1920x1080 256 spp", w, "scenes/weekend-final.json", "bvh", 1920, 256)
w.close()
for algo, gs in (("bvh", None), ("grid", 25.0), ("grid", 5.0), ("grid", 1.0)):
    w = HittableList.from_ply("scenes/dragon_recon/dragon_vrip_res4.ply")
    run("configs[2] dragon res4 900x506 50 spp", w, "scenes/dragon_recon/dragon_vrip_res4.ply", algo, 900, 50, gs)
    w.close()
w = HittableList.from_json("scenes/weekend-final.json"); w.append(HittableList.from_json("scenes/triangle_materials.json"))
run("configs[4] weekend-final + triangle_materials BVH 900x506 4096 spp", w, "scenes/weekend-mixed.json", "bvh", 900, 4096, repeat=1)
w.close()
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/configs.json", "w"), indent=1)
