#!/usr/bin/env python3
"""Run ON THE GPU BOX: the north_star's image check at its stated size — GPU render vs the CPU oracle with an INDEPENDENT seed at
480x270 and many samples per pixel (>= 40 dB needs both sides converged).  Writes gpurun_out/psnr_evidence.json.
usage: tools/psnr_evidence.py [scene spp]..."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import raytracer_amd as rt
from raytracer_amd import Camera, HittableList
from oracle import oracle_py as O
rt.init(0)
args = sys.argv[1:] or ["weekend-final", "16384", "dragon", "2048"]
out = []
for name, spp in zip(args[0::2], map(int, args[1::2])):
    if name == "dragon":
        w = HittableList.from_ply("scenes/dragon_recon/dragon_vrip_res3.ply", subdivide=2); fn = "scenes/dragon_recon/dragon_vrip_res3.ply"
    else:
        fn = f"scenes/{name}.json"; w = HittableList.from_json(fn)
    w.algorithm = "bvh"; w.init(); w.upload()
    cam = Camera(fn, image_width=480, samples_per_pixel=spp)
    t = time.time(); img, st = cam.render(w, seed=11); t_gpu = time.time() - t
    osc = O.OracleScene(w.prims(), w.centroids(), w.materials()); osc.build_bvh("sah-position", sweep=True)
    oc = O.camera_from_rt(cam.c); H = cam.image_height
    ref = np.zeros_like(img); t = time.time(); rays = 0
    for r0 in range(0, H, 10):                                   # bands of rows: progress lines for the harness
        band, c = osc.render(oc, "bvh", 0, spp, cam.max_depth, seed=22, rows=(r0, min(H, r0 + 10)), nthreads=16)
        ref += band; rays += c["primary_rays"] + c["scatter_rays"]
        print(f"{name}: CPU rows {r0 + 10}/{H}, {time.time() - t:.0f} s", flush=True)
    t_cpu = time.time() - t
    a, b = cam.resolve(img).astype(np.float64), O.resolve(ref, spp).astype(np.float64)
    mse = np.mean((a - b) ** 2); psnr = 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)
    r = {"scene": name, "image": f"480x{H}", "spp_each_side": spp, "gpu_seed": 11, "cpu_seed": 22, "psnr_db_8bit_gamma": psnr,
         "rmse_linear": float(np.sqrt(np.mean((img / spp - ref / spp) ** 2))), "gpu_seconds": t_gpu, "cpu_seconds_16_threads": t_cpu,
         "gpu_Mrays_per_s": (st["primary_rays"] + st["scatter_rays"]) / t_gpu / 1e6, "cpu_Mrays_per_s_16_threads": rays / t_cpu / 1e6}
    out.append(r); print(json.dumps(r), flush=True)
    w.close()
os.makedirs("gpurun_out", exist_ok=True); json.dump(out, open("gpurun_out/psnr_evidence.json", "w"), indent=1)
