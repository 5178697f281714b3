"""world_size-2 `gloo` test of the N>1 path on CPU: spp-split + one reduce of the f64 framebuffers.
The per-rank render is stood in for by the oracle (no GPU here); what is under test is the host logic bench.py uses:
raytracer_amd.dist.spp_range and reduce_framebuffer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd.dist import reduce_framebuffer, spp_range


def test_spp_range_partitions_exactly():
    for world in (1, 2, 3, 4, 8):
        for spp in (1, 7, 50, 128, 1024):
            got = [spp_range(r, world, spp) for r in range(world)]
            covered = [s for b, c in got for s in range(b, b + c)]
            assert covered == list(range(spp))
    with pytest.raises(ValueError):
        spp_range(2, 2, 8)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import oracle_py as O
    from raytracer_amd import Camera, HittableList
    w = HittableList.from_json("scenes/simple_hollow_glass.json")
    osc = O.OracleScene(w.prims(), w.centroids(), w.materials()).build_bvh("sah-position")
    cam = Camera("scenes/simple_hollow_glass.json", image_width=48, samples_per_pixel=6)
    b, c = spp_range(rank, world, 6)
    img, st = osc.render(O.camera_from_rt(cam.c), "bvh", b, c, 50, seed=4)
    fb = torch.from_numpy(img.copy())
    reduce_framebuffer(fb, dst=0)
    rays = torch.tensor([float(st["primary_rays"] + st["scatter_rays"])], dtype=torch.float64)
    dist.all_reduce(rays)
    if rank == 0:
        full, fst = osc.render(O.camera_from_rt(cam.c), "bvh", 0, 6, 50, seed=4)
        q.put((np.allclose(fb.numpy(), full, rtol=1e-13, atol=0), float(rays.item()) == float(fst["primary_rays"] + fst["scatter_rays"])))
    dist.barrier()
    dist.destroy_process_group()


def test_spp_split_reduce_world_size_2():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    for p in procs: p.join(120)
    assert all(p.exitcode == 0 for p in procs)
    same_image, same_rays = q.get()
    assert same_image and same_rays


def test_bench_gpus_flag_starts_that_many_ranks(monkeypatch):
    """bench.py --gpus N without a launcher builds a torch.distributed.run command for N ranks on 127.0.0.1 (and does so
    before importing torch); under a launcher a WORLD_SIZE that contradicts --gpus is an error."""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "2"]
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.main()


def test_multi_gpu_bench_line_reuses_the_committed_single_gpu_counters(tmp_path):
    """N > 1 cannot run the profiler passes; rank 0 traces the paths of the N = 1 run, so bench.py prices its own times with the
    committed N = 1 counters when the per-GPU workload AND the device sources (by hash) are the same — flagged measured = false —
    and with nothing otherwise (a kernel change must not silently report a stale VALU fraction)."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test2", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    ref = json.load(open(os.path.join(root, "profiles", "round2_bench.json")))     # a real N = 1 line (round 2's; it carries no hash)
    ref["source_hash"] = bench.source_hash()
    line = tmp_path / "committed.json"; line.write_text(json.dumps(ref))
    bench.COMMITTED_LINE = str(line)

    class A:
        spp_per_gpu, max_depth = 128, 50
    wl8 = ref["config"]["workload"].replace("(128 spp total", "(1024 spp total")
    p = bench.committed_pmc(wl8, 8, A)
    assert p and p["committed"] and "committed N = 1 run" in p["source"] and p["child"]["segments"] == ref["roofline"]["pmc"]["segments_profiled"]
    r = ref["roofline"]
    seg = p["child"]["segments"]
    c = {"segments": seg, "box_tests": int(r["box_tests_per_segment"] * seg), "prim_tests": int(r["prim_tests_per_segment"] * seg),
         "launches": 45, "extend_s": r["avg_launch_ms"] * 45 / 1e3, "pair_bytes": 64, "prim_bytes": 48}
    rr = bench.roofline(c, p)
    assert rr["bound"] == "valu" and rr["measured"] is False and abs(rr["frac"] - r["frac"]) < 0.01 and 0 < rr["hbm"]["frac"] < 1
    A.spp_per_gpu = 64
    assert bench.committed_pmc(wl8, 8, A) is None
    A.spp_per_gpu = 128
    ref["source_hash"] = "0" * 16; line.write_text(json.dumps(ref))        # counters of other kernels: not used
    assert bench.committed_pmc(wl8, 8, A) is None
    rr = bench.roofline(c, None)
    assert rr["bound"] == "l1_lane_access" and rr["measured"] is True
    # the line the repository commits for this round, when present, is one bench.py can read
    if os.path.exists(os.path.join(root, "profiles", "round3_bench.json")):
        cur = json.load(open(os.path.join(root, "profiles", "round3_bench.json")))
        assert "source_hash" in cur and "counters" in cur["roofline"]["pmc"]
