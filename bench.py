#!/usr/bin/env python3
"""bench.py — headline benchmark: Mrays/s of the ray_color path loop, dragon PLY, BVH (SAH), MI355X.

Workload (BASELINE.json configs[3]): full-resolution Stanford dragon, `--algorithm bvh` (SAH position),
3840x2160, max depth 50, 1024 spp split over 8 GPUs = **128 spp per GPU** (weak scaling: every rank renders
its own 128 global sample indices of every pixel with a replicated scene).  The full-resolution
`dragon_vrip.ply` is absent from the reference checkout (.MISSING_LARGE_BLOBS); when it is not found under
scenes/dragon_recon/ the mesh is the committed res3 dragon with every triangle split into 16 (764 704
triangles, labelled "synthetic" in the output).

One step = one pass of the hot path over the rank's batch: generate -> (extend, shade)* -> accumulate for
128 spp of the 3840x2160 frame, then (N > 1) one RCCL reduce of the f64 framebuffers to rank 0.
Timed region starts with the scene, BVH and queues resident in HBM.
Metric: Mrays/s = (primary_rays + scatter_rays) / seconds / 1e6, whole job (src/camera.rs:70,171).

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process starts N ranks itself (torch.distributed.run,
one per GPU) before it touches torch or the GPU, and relays rank 0's JSON line.  Under torchrun the ranks check that
WORLD_SIZE equals --gpus.

Roofline (N = 1): after the timed steps the workspace is released and the same workload is run once more under
`rocprofv3 --pmc` in child processes (one pass per counter group), so the VALU and fabric-traffic figures on the line come
from counters of this run's own kernels; `--no-pmc` skips that and the line says so.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import signal
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.chdir(ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--spp-per-gpu", type=int, default=128)
    ap.add_argument("--max-depth", type=int, default=50)
    ap.add_argument("--scene", default="dragon", choices=["dragon", "dragon_res3", "dragon_res4", "weekend-final", "weekend-mixed"])
    ap.add_argument("--algorithm", default="bvh", choices=["bvh", "grid", "naive"], help="profiling of the other walks; the headline is bvh")
    ap.add_argument("--grid-size", type=float, default=5.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL); gloo is for rehearsing N>1 on a one-GPU box")
    ap.add_argument("--force-device", type=int, default=-1, help="rehearsal only: every rank uses this GPU (needs --backend gloo)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target CPU-baseline duration")
    ap.add_argument("--reduce", default="torch", choices=["torch", "product"],
                    help="the per-step framebuffer reduce: torch.distributed.reduce (RCCL through torch; default) or the product library's own "
                         "communicator (rt_comm_reduce_sum = ncclReduce; also exercised at N = 1)")
    ap.add_argument("--dist-at-1", action="store_true", help="rehearsal: with --gpus 1 under a launcher, still create the process group and run every "
                    "collective of the N > 1 path (init, warm-up all-reduce, the framebuffer reduce, barrier, the MAX / SUM all-reduces) over one rank")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes that feed the roofline")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)      # the profiled re-run: one step, counters only
    return ap.parse_args()


def launch_ranks(a):
    """--gpus N without a launcher: start N ranks (one process per GPU) from a parent that never touches the GPU."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "16")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def load_scene(name):
    from raytracer_amd import HittableList
    full = "scenes/dragon_recon/dragon_vrip.ply"
    if name == "dragon":
        if os.path.exists(full):
            return HittableList.from_ply(full), full, "dragon_vrip.ply (871k tris)"
        w = HittableList.from_ply("scenes/dragon_recon/dragon_vrip_res3.ply", subdivide=2)
        return w, "scenes/dragon_recon/dragon_vrip_res3.ply", "synthetic: dragon_vrip_res3.ply, each triangle split into 16 (764704 tris); dragon_vrip.ply absent"
    if name == "dragon_res3":
        p = "scenes/dragon_recon/dragon_vrip_res3.ply"; return HittableList.from_ply(p), p, "dragon_vrip_res3.ply (47794 tris)"
    if name == "dragon_res4":
        p = "scenes/dragon_recon/dragon_vrip_res4.ply"; return HittableList.from_ply(p), p, "dragon_vrip_res4.ply (11102 tris)"
    if name == "weekend-final":
        p = "scenes/weekend-final.json"; return HittableList.from_json(p), p, "weekend-final.json (487 spheres)"
    w = HittableList.from_json("scenes/weekend-final.json"); w.append(HittableList.from_json("scenes/triangle_materials.json"))
    return w, "scenes/weekend-mixed.json", "weekend-final + triangle_materials (493 objects)"


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_ranks(a))
    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU, or let --gpus start them")
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if a.force_device >= 0:
        local_rank = a.force_device
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or a.dist_at_1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    import raytracer_amd as rt
    from raytracer_amd import Camera
    rt.init(local_rank)
    rt.set_profiling(a.pmc_child)                             # the timed steps run the path and nothing else (no event records)

    t0 = time.time()
    w, cam_name, label = load_scene(a.scene)
    w.algorithm, w.bvh_mode, w.grid_size = a.algorithm, "sah-position", a.grid_size
    w.init()
    w.upload()
    init_s = time.time() - t0
    cam = Camera(cam_name, image_width=a.width, samples_per_pixel=a.spp_per_gpu * world, max_depth=a.max_depth)
    H, W = cam.image_height, cam.image_width
    fb = torch.zeros((H, W, 3), dtype=torch.float64, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    from raytracer_amd.dist import reduce_framebuffer, spp_range
    s_begin, s_count = spp_range(rank, world, a.spp_per_gpu * world)

    # set-up that is not a step: the render workspace (one hipMalloc set of ~200 GB at the default size) and the communicator's
    # first collective, so that a run with --warmup 0 times steps and nothing else
    cam.reserve(w, s_count)
    if use_dist:
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device="cuda"))
    torch.cuda.synchronize()

    comm = None
    if a.reduce == "product":                                 # the library's own RCCL communicator (include/rt_hip.h rt_comm_*)
        ids = [rt.Comm.unique_id() if rank == 0 else None]
        if use_dist:
            dist.broadcast_object_list(ids, src=0)
        comm = rt.Comm(ids[0], world, rank)

    def step():
        _, st = cam.render(w, seed=1, s_begin=s_begin, s_count=s_count, d_out=fb.data_ptr(), stream=stream)
        if comm is not None:
            comm.reduce_sum(fb.data_ptr(), fb.numel(), root=0, stream=stream)
        else:
            reduce_framebuffer(fb, dst=0, force=a.dist_at_1)    # RCCL over xGMI: the one exchange step of the path
        return st

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        print(f"[bench] set-up done in {time.time() - t0:.1f} s; {a.warmup} warm-up + {a.steps} timed steps ...", file=sys.stderr, flush=True)
    for _ in range(a.warmup):
        step()
    fence()
    t1 = time.perf_counter()
    agg = None
    for _ in range(a.steps):
        st = step()
        if agg is None:
            agg = dict(st)
        else:
            for k in st:
                agg[k] += st[k]
    fence()
    dt = time.perf_counter() - t1
    # one more step with per-kernel HIP events on the library's launch stream (rt_set_profiling): the kernel times the roofline
    # divides by.  Same seed, same paths: its counters equal one timed step's.
    if a.pmc_child:                                           # the counter passes: exactly one step under the profiler
        prof, dt_prof = st, dt
    else:
        rt.set_profiling(True)
        fence()
        t2 = time.perf_counter()
        prof = step()
        fence()
        dt_prof = time.perf_counter() - t2
        rt.set_profiling(False)
    tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
    rays = torch.tensor([float(agg["primary_rays"] + agg["scatter_rays"])], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    dt = float(tt.item()); total_rays = float(rays.item())

    if rank == 0:
        seg, nbox, nprim = prof["ray_segments"], prof["box_tests"], prof["prim_tests"]
        launches = max(prof["extend_launches"], 1)
        ext_s = prof["extend_ms"] / 1e3
        pair_b, prim_b = w.bvh_layout() if a.algorithm == "bvh" else (0, 80)   # what the kernel's layout for this scene really reads
        counts = {"segments": seg, "box_tests": nbox, "prim_tests": nprim, "launches": launches, "extend_s": ext_s,
                  "pair_bytes": pair_b, "prim_bytes": prim_b}
        out = {
            "metric": "Mrays/s (whole node) dragon PLY BVH; PSNR vs CPU render checked in tests",
            "value": total_rays / dt / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if "synthetic" in label else "real mesh",
            "config": {"workload": f"{label}; --algorithm {'bvh (SAH position)' if a.algorithm == 'bvh' else a.algorithm + (f' size={a.grid_size:g}' if a.algorithm == 'grid' else '')}; {W}x{H}; {a.spp_per_gpu} spp per GPU "
                                   f"({a.spp_per_gpu * world} spp total, spp-split); max_depth {a.max_depth}; camera preset dragon/0",
                       "primitives": len(w), "bvh_nodes": w.bvh.n_nodes if w.bvh else None, "bvh_depth": w.bvh.depth if w.bvh else None,
                       "parallelism": (f"spp-split x{world}, replicated scene, one RCCL reduce of the f64 framebuffer per step ({world} ranks, "
                                       f"{'rt_comm_reduce_sum of the product library' if a.reduce == 'product' else 'torch.distributed.reduce'})") if world > 1 else "single GPU",
                       "init_seconds_excluded": round(init_s, 2)},
            "rays_per_step": total_rays / a.steps,
            "kernel_ms_per_step": {k: prof[k] for k in ("generate_ms", "extend_ms", "shade_ms", "resolve_ms")},
            "profiled_step": {"ms": dt_prof * 1e3, "note": "one extra step after the timed ones with per-kernel HIP events on; "
                              "kernel_ms_per_step and the roofline's launch times come from it, value / ms_per_step from the un-profiled steps"},
            "source_hash": source_hash(),
        }
        if a.pmc_child:                                         # the profiled re-run only reports what the parent normalises by
            out["pmc_child"] = counts
            print(json.dumps(out), flush=True)
            return
        pmc = None
        if world == 1 and not a.no_pmc:
            # free this process's HBM (the workspace alone is ~200 GB) before the profiled children allocate theirs
            w.close(); del fb; torch.cuda.empty_cache()
            pmc = pmc_passes(a)
        if pmc is None and world > 1:
            pmc = committed_pmc(out["config"]["workload"], world, a)
        out["roofline"] = roofline(counts, pmc)
        if world == 1 and not a.no_cpu_baseline:
            print("[bench] CPU baseline (oracle, 1 core, then all cores) ...", file=sys.stderr, flush=True)
            out["cpu_baseline"] = cpu_baseline(w, cam_name, a)
            out["speedup_vs_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"] if out["cpu_baseline"]["value"] else None
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


# ---- roofline of the dominant kernel (k_extend_bvh2) ----------------------------------------------------------------
# Hardware figures (MI355X_MICROARCH.md): HBM3E 8 TB/s; vector f64 78.6 TFLOP/s = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x
# 2.4 GHz, i.e. 3.93e13 VALU lane-operations per second at full rate (every f64 and integer instruction of this kernel
# issues at that rate: tools/mb/mb_valu.hip; only rcp/rsq are slower and the traversal loop has none).
VALU_PEAK_LANE_OPS = 256 * 64 * 2.4e9
HBM_PEAK_GBPS = 8000.0
PMC_PASSES = {
    "sq": "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE",
    "fetch": "FETCH_SIZE",
    "write": "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum",
    # the vector-memory address path (texture addresser + vector L1) of the CUs: what the extend kernel's four divergent 16-byte lane
    # accesses per visit load (profiles/round3_default_ta_tcp.md).  Optional: a failure of this pass leaves the rest standing
    "ta": "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum",
}
PMC_OPTIONAL = {"ta"}


def fetch_calibration():
    """FETCH_SIZE bytes reported per byte really fetched, for this kernel's access shape (per-lane 16-B loads of 64-B records
    at unrelated addresses), measured by tools/mb/mb_fetch.hip under the same counter: profiles/round2_fetch_calibration.json.
    The guide's x2 rule is for wide streaming reads only."""
    try:
        c = json.load(open(os.path.join(ROOT, "profiles", "round2_fetch_calibration.json")))
        return float(c["bytes_fetched_per_byte_reported"]), c.get("source", "profiles/round2_fetch_calibration.json")
    except Exception:
        return 1.0, "uncalibrated (FETCH_SIZE taken as reported)"


def pmc_passes(a):
    """One `rocprofv3 --pmc <group> -- python3 bench.py --pmc-child ...` per counter group: the same workload, one step.
    Returns {"counters": {name: sum over the extend dispatches}, "child": counts of the profiled step, ...} or a dict with
    "error".  Each pass is a fresh process (the profiler's library initialises the GPU before the program starts)."""
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="rt_pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    keep = os.path.join(ROOT, "gpurun_out", "bench_pmc")
    child = [sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-pmc",
             "--width", str(a.width), "--spp-per-gpu", str(a.spp_per_gpu), "--max-depth", str(a.max_depth), "--scene", a.scene,
             "--algorithm", a.algorithm, "--grid-size", str(a.grid_size)]
    res = {"counters": {}, "dispatches": 0, "child": None, "seconds": {}, "kernel_ns": 0.0,
           "source": "rocprofv3 --pmc passes of this run's own workload (one step per pass, separate child processes)"}
    env = dict(os.environ); env["TMPDIR"] = "/tmp"
    for name, ctrs in PMC_PASSES.items():
        d = os.path.join(tmp, name)
        cmd = [prof, "--pmc"] + ctrs.split() + ["--output-format", "csv", "-d", d, "-o", name, "--"] + child
        t0 = time.time()
        print(f"[bench] rocprofv3 --pmc pass '{name}' ({ctrs}) ...", file=sys.stderr, flush=True)
        try:
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, start_new_session=True)
            try:
                log, _ = p.communicate(timeout=120)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL); p.wait()
                if name in PMC_OPTIONAL:
                    res["optional_error"] = f"pass {name}: timed out"; continue
                res["error"] = f"pass {name}: timed out"; break
        except OSError as e:
            res["error"] = f"pass {name}: {e}"; break
        res["seconds"][name] = round(time.time() - t0, 1)
        lines = [l for l in log.splitlines() if l.startswith('{"metric"')]
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not lines or not f:
            if name in PMC_OPTIONAL:
                res["optional_error"] = f"pass {name}: rc {p.returncode}; " + log[-300:].replace("\n", " | "); continue
            res["error"] = f"pass {name}: rc {p.returncode}; " + log[-300:].replace("\n", " | "); break
        if res["child"] is None:
            res["child"] = json.loads(lines[-1])["pmc_child"]
        seen = set()
        for r in csv.DictReader(open(f[0])):
            if "k_extend" not in r["Kernel_Name"]:
                continue
            res["counters"][r["Counter_Name"]] = res["counters"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if name == "sq" and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); res["kernel_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if name == "sq":
            res["dispatches"] = len(seen)
        try:                                                   # keep the raw csv where a gpurun call merges it back
            os.makedirs(keep, exist_ok=True); shutil.copy(f[0], os.path.join(keep, f"{name}_counter_collection.csv"))
        except OSError:
            pass
    shutil.rmtree(tmp, ignore_errors=True)
    return res


def source_hash():
    """sha1 over the device sources the kernels are built from: a committed N = 1 line is only reused (committed_pmc) for the
    kernels it was measured on."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "raytracer_amd", "csrc", "device", "*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


COMMITTED_LINE = os.path.join("profiles", "round3_bench.json")


def committed_pmc(workload, world, a):
    """N > 1: rank 0 renders global samples [0, spp_per_gpu) of every pixel - exactly the paths of the N = 1 run, so its extend
    kernel does the same work per ray segment.  The counters of the committed N = 1 line (same scene / image / spp per GPU /
    depth, and the same device sources by hash) are used with this run's own times; the roofline then says measured = false.
    A line measured on other kernels is not used: the counter-free lane-access fraction is reported instead."""
    try:
        ref = json.load(open(os.path.join(ROOT, COMMITTED_LINE)))
        same = ref["config"]["workload"].split(";")[:3] == workload.split(";")[:3] and f"{a.spp_per_gpu} spp per GPU" in ref["config"]["workload"] \
            and f"max_depth {a.max_depth}" in ref["config"]["workload"] and ref.get("source_hash") == source_hash()
        p = ref["roofline"]["pmc"]
        if not same or "counters" not in p:
            return None
        return {"counters": p["counters"], "child": {"segments": p["segments_profiled"]}, "dispatches": p["dispatches_profiled"], "seconds": {}, "kernel_ns": 0.0,
                "committed": True,
                "source": f"counters of the committed N = 1 run of the same per-GPU workload and the same device sources ({COMMITTED_LINE}), scaled by ray "
                          f"segments to rank 0's profiled step of this {world}-GPU run: rank 0 traces the same paths"}
    except Exception:
        return None


def roofline(c, pmc):
    seg, nbox, nprim, launches, ext_s = c["segments"], c["box_tests"], c["prim_tests"], c["launches"], c["extend_s"]
    pair_b, prim_b = c["pair_bytes"], c["prim_bytes"]
    alg_bytes = 60.0 * seg + 64.0 * nbox + 72.0 * nprim          # (48 in + 12 out) + 64 per box test + 72 per triangle test
    layout_bytes = 60.0 * seg + 0.5 * pair_b * nbox + prim_b * nprim
    lane_acc = ((pair_b / 16.0) * 0.5 * nbox + (prim_b / 16.0) * nprim + 6.0 * seg) / ext_s / (256 * 2.4e9) if ext_s > 0 else 0.0
    r = {"kernel": "k_extend_bvh2" if pair_b else "k_extend_grid / k_extend_naive", "avg_launch_ms": ext_s * 1e3 / launches, "launches": launches,
         # the contract's byte model, as plain fields: it prices every box test at 64 B from HBM, while the tree sits in the
         # Infinity Cache / L2 and the compact layout requests half of it, so this rate can exceed the HBM peak and is no bound
         "algorithmic_bytes_per_launch": alg_bytes / launches, "algorithmic_GBps": alg_bytes / ext_s / 1e9 if ext_s > 0 else 0.0,
         "bytes_per_segment": alg_bytes / max(seg, 1), "box_tests_per_segment": nbox / max(seg, 1), "prim_tests_per_segment": nprim / max(seg, 1),
         "layout": {"pair_bytes": pair_b, "prim_bytes": prim_b, "requested_GBps": layout_bytes / ext_s / 1e9 if ext_s > 0 else 0.0},
         # secondary, against a self-measured peak: divergent <=16-B lane accesses per clock per CU vs the 1.1 of a pure gather
         # (tools/mb/mb_gather.hip, profiles/round1_mb_gather.txt)
         "l1_lane_access_bound": {"achieved": lane_acc, "peak": 1.1, "unit": "lane accesses/clk/CU", "frac": lane_acc / 1.1}}
    k = (pmc or {}).get("counters") or {}
    ch = (pmc or {}).get("child") or {}
    need = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU")
    if pmc and "error" not in pmc and all(k.get(n) for n in need) and ch.get("segments"):
        # counters are sums over the extend dispatches of ONE profiled step; the work per segment is deterministic (same seed,
        # same paths), so they are scaled by segments to the timed steps and divided by the un-profiled HIP-event time
        scale = seg / ch["segments"]
        lane_util = k["SQ_THREAD_CYCLES_VALU"] / (k["SQ_ACTIVE_INST_VALU"] * 64.0)
        useful = k["SQ_INSTS_VALU"] * 64.0 * lane_util * scale     # lane-operations that did work
        issued = k["SQ_INSTS_VALU"] * 64.0 * scale                 # lane slots the issued instructions occupied
        achieved = useful / ext_s
        r.update({"measured": not pmc.get("committed", False),   # false: counters of the committed N = 1 line, not of this run
                  "bound": "valu", "achieved": achieved / 1e12, "peak": VALU_PEAK_LANE_OPS / 1e12, "unit": "T lane-op/s (f64-rate VALU)",
                  "frac": achieved / VALU_PEAK_LANE_OPS,
                  "valu": {"lane_utilisation": lane_util, "issue_frac": issued / ext_s / VALU_PEAK_LANE_OPS,
                           "valu_wave_insts_per_segment": k["SQ_INSTS_VALU"] / ch["segments"],
                           "wave_wait_fraction": k.get("SQ_WAIT_ANY", 0.0) / k["SQ_WAVE_CYCLES"] if k.get("SQ_WAVE_CYCLES") else None,
                           "clock_GHz_profiled": (k["GRBM_GUI_ACTIVE"] / 8.0 / pmc["kernel_ns"]) if k.get("GRBM_GUI_ACTIVE") and pmc.get("kernel_ns") else None,
                           # occupancy: wave lifetimes (SQ_WAVE_CYCLES counts quad-cycles) over the kernel's cycles, per SIMD (1024 of them);
                           # the kernel's 126 VGPRs allow 4
                           "mean_resident_waves_per_simd": (k["SQ_WAVE_CYCLES"] * 4.0 / (k["GRBM_GUI_ACTIVE"] / 8.0) / 1024.0)
                           if k.get("GRBM_GUI_ACTIVE") and k.get("SQ_WAVE_CYCLES") else None}})
        if k.get("FETCH_SIZE") is not None and k.get("WRITE_SIZE") is not None:
            cal, cal_src = fetch_calibration()
            fetch_b = k["FETCH_SIZE"] * 1024.0 * cal * scale; write_b = k["WRITE_SIZE"] * 1024.0 * scale      # counter unit: KiB
            r["traffic"] = (fetch_b + write_b) / launches
            r["hbm"] = {"bound": "hbm", "achieved": (fetch_b + write_b) / ext_s / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": (fetch_b + write_b) / ext_s / 1e9 / HBM_PEAK_GBPS,
                        "fetch_bytes_per_segment": fetch_b / seg, "write_bytes_per_segment": write_b / seg,
                        "fetch_size_calibration": cal, "calibration_source": cal_src,
                        "l2_hit_rate": k["TCC_HIT_sum"] / (k["TCC_HIT_sum"] + k["TCC_MISS_sum"]) if k.get("TCC_HIT_sum") and k.get("TCC_MISS_sum") is not None else None,
                        "note": "fabric-side bytes (L2 misses; Infinity-Cache hits are counted as fetches), so an upper bound of the HBM traffic"}
        else:
            r["traffic"] = None
        if k.get("TA_TA_BUSY_sum") and k.get("GRBM_GUI_ACTIVE"):
            cu_cycles = 256.0 * k["GRBM_GUI_ACTIVE"] / 8.0           # GRBM_GUI_ACTIVE is summed over the 8 XCDs; one TA / TCP per CU
            r["ta"] = {"busy_frac": k["TA_TA_BUSY_sum"] / cu_cycles,
                       "busy_cycles_per_wave_instruction": k["TA_TA_BUSY_sum"] / k["TA_TOTAL_WAVEFRONTS_sum"] if k.get("TA_TOTAL_WAVEFRONTS_sum") else None,
                       "tcp_cache_accesses_per_clk_per_cu": k.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0.0) / cu_cycles,
                       "tcp_pending_stall_frac": k.get("TCP_PENDING_STALL_CYCLES_sum", 0.0) / cu_cycles,
                       "l1_hit_rate": 1.0 - k["TCP_TCC_READ_REQ_sum"] / k["TCP_TOTAL_CACHE_ACCESSES_sum"] if k.get("TCP_TOTAL_CACHE_ACCESSES_sum") and k.get("TCP_TCC_READ_REQ_sum") is not None else None,
                       "note": "hardware-counter counterpart of l1_lane_access_bound: the fraction of cycles the CUs' texture addressers are busy with this kernel's loads"}
        r["pmc"] = {"source": pmc["source"], "dispatches_profiled": pmc["dispatches"], "pass_seconds": pmc["seconds"],
                    "counters": {n: v for n, v in sorted(k.items())}, "segments_profiled": ch["segments"]}
    else:
        # no counters from this run: the fraction that needs none (lane accesses against the measured gather rate) is the
        # headline, and the line says why
        r.update({"measured": True, "bound": "l1_lane_access", "achieved": lane_acc, "peak": 1.1, "unit": "lane accesses/clk/CU", "frac": lane_acc / 1.1,
                  "traffic": None, "pmc": {"source": "none", "reason": (pmc or {}).get("error", "PMC passes not run (--no-pmc or N > 1)")}})
    return r


def cpu_baseline(w, cam_name, a):
    """The CPU oracle (a restatement — kind "port" — of the single-threaded reference) timed on this host, 1 core,
    on a bounded sample: the same scene, BVH and camera at 480x270, spp chosen for ~cpu-seconds of work."""
    from oracle import oracle_py as O
    from raytracer_amd import Camera
    osc = O.OracleScene(w.prims(), w.centroids(), w.materials())
    t = time.perf_counter(); osc.build_bvh("sah-position", sweep=True); t_build = time.perf_counter() - t   # the oracle's own SAH build
    on, oo = osc.get_bvh()
    same_tree = on.tobytes() == w.bvh.nodes().tobytes() and bool((oo == w.bvh.order(len(w))).all())
    cam = Camera(cam_name, image_width=480, max_depth=a.max_depth)
    oc = O.camera_from_rt(cam.c)
    t = time.perf_counter(); _, c = osc.render(oc, "bvh", 0, 1, a.max_depth, seed=1, nthreads=1); d1 = time.perf_counter() - t
    spp = max(1, min(256, int(a.cpu_seconds / max(d1, 1e-3))))
    t = time.perf_counter(); _, c = osc.render(oc, "bvh", 1, spp, a.max_depth, seed=1, nthreads=1); dt = time.perf_counter() - t
    rays = c["primary_rays"] + c["scatter_rays"]
    # the same oracle with image rows split over host threads — NOT reference behaviour, reported beside it
    nth = max(1, min(16, int(os.environ.get("OMP_NUM_THREADS", "16")), os.cpu_count() or 1))
    t = time.perf_counter(); _, c2 = osc.render(oc, "bvh", 1, spp, a.max_depth, seed=1, nthreads=nth); dt2 = time.perf_counter() - t
    rays2 = c2["primary_rays"] + c2["scatter_rays"]
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "bvh_build_seconds": round(t_build, 2),
            "bvh_identical_to_gpu_path": same_tree,
            "sample": f"same scene/BVH/camera at 480x{cam.image_height}, {spp} spp, depth {a.max_depth}: {rays} rays in {dt:.1f} s on 1 host core "
                      f"(host has {os.cpu_count()} logical CPUs; the reference is single-threaded by construction)",
            "all_cores": {"value": rays2 / dt2 / 1e6, "unit": "Mrays/s", "cores": nth, "note": "rows split over std::threads; not reference behaviour"}}


if __name__ == "__main__":
    main()
