#!/usr/bin/env python3
"""Run ON THE GPU BOX: the north_star's bit-exactness claim at the bench's full size — every primary ray of sample 0 of the
3840x2160 frame over the 764 704-triangle mesh, GPU rt_trace vs the CPU oracle: hit flag, primitive index, rec.t and the parameter
of rec.p compared bit for bit, and the traversal counters.  Writes gpurun_out/primary_hits_evidence.json."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import raytracer_amd as rt
from raytracer_amd import Camera, HittableList
from oracle import oracle_py as O
rt.init(0)
fn = "scenes/dragon_recon/dragon_vrip_res3.ply"
w = HittableList.from_ply(fn, subdivide=2); w.algorithm = "bvh"; w.init(); w.upload()
cam = Camera(fn, image_width=3840, samples_per_pixel=1)
W, H = cam.image_width, cam.image_height
osc = O.OracleScene(w.prims(), w.centroids(), w.materials()); osc.build_bvh("sah-position", sweep=True)
tot = {"rays": 0, "hits": 0, "mismatch_hit": 0, "mismatch_prim": 0, "mismatch_t_bits": 0, "mismatch_tp_bits": 0, "mismatch_ray_bits": 0}
cnt_g = {}; cnt_o = {}; t0 = time.time()
for j0 in range(0, H, 120):                                    # bands of rows: bounded memory, progress lines
    jj, ii = np.meshgrid(np.arange(j0, min(H, j0 + 120), dtype=np.uint32), np.arange(W, dtype=np.uint32), indexing="ij")
    o, d = cam.get_rays(ii.ravel(), jj.ravel(), sample=0, seed=1)
    oo, od = O.generate(O.camera_from_rt(cam.c), 1, 0, ii.ravel(), jj.ravel())
    tot["mismatch_ray_bits"] += int((o.view(np.uint64) != oo.view(np.uint64)).sum() + (d.view(np.uint64) != od.view(np.uint64)).sum())
    g = w.hit(o, d); r = osc.trace("bvh", o, d)
    tot["rays"] += len(o); tot["hits"] += int(r["hit"].sum())
    tot["mismatch_hit"] += int((g["hit"] != r["hit"]).sum()); tot["mismatch_prim"] += int((g["prim"] != r["prim"]).sum())
    tot["mismatch_t_bits"] += int((g["t"].view(np.uint64) != r["t"].view(np.uint64)).sum())
    tot["mismatch_tp_bits"] += int((g["tp"].view(np.uint64) != r["tp"].view(np.uint64)).sum())
    for k in ("intersection_checks", "traversal_steps", "overlapping_aabb"):
        cnt_g[k] = cnt_g.get(k, 0) + int(g["stats"][k]); cnt_o[k] = cnt_o.get(k, 0) + int(r["stats"][k])
    print(f"rows {min(H, j0 + 120)}/{H}  {time.time() - t0:.0f} s  mismatches so far: {sum(v for k, v in tot.items() if k.startswith('mismatch'))}", flush=True)
out = {"workload": f"{W}x{H} primary rays (sample 0, seed 1) over {len(w)} triangles, BVH SAH-position", **tot, "counters_gpu": cnt_g, "counters_oracle": cnt_o,
       "counters_equal": cnt_g == cnt_o}
os.makedirs("gpurun_out", exist_ok=True); json.dump(out, open("gpurun_out/primary_hits_evidence.json", "w"), indent=1)
print(json.dumps(out))
