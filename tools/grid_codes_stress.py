#!/usr/bin/env python3
"""Run ON THE GPU BOX: differential test of the grid walk's cell codes and exact integer stepping (grid_cell_box / grid_step_regular,
rt_kernels.hip) against the CPU oracle's literal Grid::traverse, at every integer cell size that tiles the 200^3 world plus sizes that
do not (codes are then off): rays from inside and outside the world, on cell planes and world faces (integer coordinates), axis-aligned
and zero direction components, plus the degenerate set of tests/test_gpu_parity.py.  usage: tools/grid_codes_stress.py
Writes gpurun_out/grid_codes_stress.json."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); os.chdir(ROOT)
import raytracer_amd as rt
from tests.test_gpu_parity import _edge_rays
from tests.util import bits, load_world, setup_pair
rt.init(0)
out = {"cases": [], "rays": 0, "mismatches": 0, "counter_mismatches": 0}
t0 = time.time()
JOBS = [("dragon_res4", [2.0, 4.0, 5.0, 8.0, 10.0, 20.0, 25.0, 40.0, 50.0, 100.0, 200.0, 3.0, 7.0, 2.5]),
        ("weekend-5-80-15", [1.0, 2.0, 5.0, 25.0, 100.0, 3.0]), ("triangle_materials", [1.0, 4.0, 50.0, 200.0, 6.0])]
for scene, sizes in JOBS:
    for gs in sizes:
        w = load_world(scene)
        try:
            osc = setup_pair(w, "grid", "sah-position", gs)
        except rt.RtError as e:                                # sizes the reference panics on (index out of bounds): an error here too
            out["cases"].append({"scene": scene, "grid_size": gs, "refused": str(e)}); print(f"{scene} size {gs:g}: refused ({e})", flush=True)
            continue
        w.upload()
        rng = np.random.default_rng(int(gs * 10) + len(scene))
        n = 6000
        o_in = rng.uniform(-100, 100, size=(n, 3)); d_in = rng.normal(size=(n, 3))
        o_out = rng.uniform(-260, 260, size=(n, 3)); d_out = -o_out + rng.normal(scale=40.0, size=(n, 3))      # towards the world from outside
        o_lat = np.round(rng.uniform(-100, 100, size=(n, 3)) / gs) * gs                                       # on cell planes / corners
        o_lat[: n // 4] = np.where(rng.random((n // 4, 3)) < 0.5, 100.0, -100.0) * (rng.random((n // 4, 3)) < 0.6) + o_lat[: n // 4] * (rng.random((n // 4, 3)) < 0.4)
        d_lat = rng.normal(size=(n, 3)); z = rng.random((n, 3)) < 0.4; d_lat[z] = rng.choice([0.0, -0.0], size=int(z.sum()))
        d_ax = np.zeros((600, 3)); d_ax[np.arange(600), rng.integers(0, 3, 600)] = rng.choice([1.0, -1.0, 3.5, -0.25], 600)
        o_ax = np.round(rng.uniform(-100, 100, size=(600, 3)))
        o_e, d_e = _edge_rays(w, 1500, int(gs) + 3)
        o = np.concatenate([o_in, o_out, o_lat, o_ax, o_e]); d = np.concatenate([d_in, d_out, d_lat, d_ax, d_e])
        bad = cbad = 0
        with np.errstate(all="ignore"):
            for tmax in (float("inf"), 60.0):
                g = w.hit(o, d, 0.001, tmax); r = osc.trace("grid", o, d, 0.001, tmax)
                m = (g["hit"] != r["hit"]) | (g["prim"] != r["prim"]) | (bits(g["t"]) != bits(r["t"])) | (bits(g["tp"]) != bits(r["tp"]))
                bad += int(m.sum())
                for k in ("intersection_checks", "traversal_steps", "gridbox_intersection_checks", "box_tests", "prim_tests"):
                    cbad += int(g["stats"][k] != r["stats"][k])
        out["rays"] += 2 * len(o); out["mismatches"] += bad; out["counter_mismatches"] += cbad
        out["cases"].append({"scene": scene, "grid_size": gs, "cells": int(w.grid.n_cells), "rays": 2 * len(o), "mismatches": bad, "counter_mismatches": cbad})
        print(f"{scene} size {gs:g}: {w.grid.n_cells} cells, mismatches {bad}, counters {cbad}", flush=True)
        w.close()
out["seconds"] = round(time.time() - t0, 1)
os.makedirs("gpurun_out", exist_ok=True); json.dump(out, open("gpurun_out/grid_codes_stress.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "cases"}))
sys.exit(1 if out["mismatches"] or out["counter_mismatches"] else 0)
