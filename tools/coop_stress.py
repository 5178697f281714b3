#!/usr/bin/env python3
"""Run ON THE GPU BOX: randomized differential test of the wave-cooperative grid cell test (k_extend_grid_coop) against the CPU
oracle's literal GridBox::hit, beyond the fixed cases of tests/test_gpu_parity.py: N triangle soups of random size / spread / grid
size / stacking, each traced with coherent fans, scattered and degenerate rays at two tmax values.  Compares hit flag, primitive,
rec.t bits, rec.p parameter bits and all counters.  usage: tools/coop_stress.py [n_scenes=40] [first_seed=100]
Writes gpurun_out/coop_stress.json."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
os.environ["RT_GRID_COOP_SHARE"] = "0"; os.environ["RT_GRID_COOP_MIN"] = "64"
import raytracer_amd as rt
from tests.test_gpu_parity import _edge_rays, _triangle_soup
from tests.util import bits, setup_pair
rt.init(0)
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
out = {"scenes": 0, "rays": 0, "chunked_cells": 0, "max_list": 0, "mismatches": 0, "counter_mismatches": 0, "cases": []}
t0 = time.time()
for s in range(seed0, seed0 + n_scenes):
    rng = np.random.default_rng(s)
    n = int(rng.choice([64, 65, 127, 128, 129, 191, 192, 193, 400, 900, 2500]))
    gs = float(rng.choice([25.0, 10.0, 5.0]))
    spread = float(rng.choice([0.3, 0.8, 2.0, 5.0]))
    ctr = tuple(float(x) for x in (rng.uniform(-20, 20, size=2).tolist() + [rng.uniform(-30, -5)]))
    w = _triangle_soup(s, n, spread, stacked=max(2, n // 40), center=ctr)
    osc = setup_pair(w, "grid", "sah-position", gs)
    ln = np.diff(w.grid.cell_offset().astype(np.int64))
    if not (ln >= 64).any():
        continue
    w.upload()
    c = np.array(ctr)
    o1 = np.tile(c + np.array([[0.0, 0.0, 9.0]]), (2048, 1)); d1 = np.array([0.0, 0.0, -1.0]) + rng.normal(scale=0.3, size=(2048, 3))
    o1b = np.tile(c + np.array([[0.0, 0.0, -9.0]]), (2048, 1)); d1b = np.array([0.0, 0.0, 1.0]) + rng.normal(scale=0.3, size=(2048, 3))
    o2 = rng.uniform(-7, 7, size=(4000, 3)) + c; d2 = rng.normal(size=(4000, 3))
    o3, d3 = _edge_rays(w, 1000, s + 7)
    o = np.concatenate([o1, o1b, o2, o3]); d = np.concatenate([d1, d1b, d2, d3])
    bad = cbad = 0
    with np.errstate(all="ignore"):
        for tmax in (float("inf"), 4.0):
            g = w.hit(o, d, 0.001, tmax); r = osc.trace("grid", o, d, 0.001, tmax)
            m = (g["hit"] != r["hit"]) | (g["prim"] != r["prim"]) | (bits(g["t"]) != bits(r["t"])) | (bits(g["tp"]) != bits(r["tp"]))
            bad += int(m.sum())
            for k in ("intersection_checks", "traversal_steps", "gridbox_intersection_checks", "box_tests", "prim_tests"):
                cbad += int(g["stats"][k] != r["stats"][k])
    out["scenes"] += 1; out["rays"] += 2 * len(o); out["chunked_cells"] += int((ln >= 64).sum()); out["max_list"] = max(out["max_list"], int(ln.max()))
    out["mismatches"] += bad; out["counter_mismatches"] += cbad
    out["cases"].append({"seed": s, "triangles": n, "grid_size": gs, "spread": spread, "chunked_cells": int((ln >= 64).sum()), "max_list": int(ln.max()), "mismatches": bad, "counter_mismatches": cbad})
    print(f"seed {s}: n={n} size={gs:g} chunked cells {(ln >= 64).sum()} max list {ln.max()} mismatches {bad} counters {cbad}", flush=True)
    w.close()
out["seconds"] = round(time.time() - t0, 1)
os.makedirs("gpurun_out", exist_ok=True); json.dump(out, open("gpurun_out/coop_stress.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "cases"}))
sys.exit(1 if out["mismatches"] or out["counter_mismatches"] else 0)
