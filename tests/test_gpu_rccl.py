"""RCCL in the product (`ncclReduce(sum, root 0)` of the accumulation framebuffers), as far as one GPU allows: a real
communicator with ONE rank on the box's GPU — rt_comm_unique_id, rt_comm_create (ncclCommInitRank), rt_comm_reduce_sum
(ncclReduce, in place) and rt_render_reduce through the C ABI, and `raytracer --gpus 1 --reduce rccl` through the CLI, whose
worker takes exactly the calls an N-GPU run takes.  A GPU can be in a communicator only once, so N > 1 over RCCL cannot run
here (tests/test_gpu_dist.py and test_gpu_cli.py rehearse N = 2, 3 with gloo / host sums); it stays unmeasured until the
driver's multi-GPU run."""
import csv
import os
import subprocess

import numpy as np
import pytest
import torch

import raytracer_amd as rt
from raytracer_amd import Camera, Comm
from raytracer_amd.host import RtError

from .util import load_world

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_reduce_and_render_reduce(gpu):
    uid = Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = Comm(uid, 1, 0)
    # ncclReduce(sum) over one rank is the identity, in place, on the caller's stream
    x = torch.arange(3 * 1000, dtype=torch.float64, device="cuda").reshape(1000, 3) * 0.25 - 7.0
    ref = x.clone()
    comm.reduce_sum(x.data_ptr(), x.numel(), root=0, stream=torch.cuda.current_stream().cuda_stream)
    assert torch.equal(x, ref)
    comm.reduce_sum(x.data_ptr(), 0)                                   # empty message: nothing to do, no error
    # rt_render_reduce == rt_render for the same samples
    w = load_world("weekend-final"); w.algorithm = "bvh"; w.init().upload()
    cam = Camera(w.filename, image_width=160, samples_per_pixel=5)
    plain, sp = cam.render(w, seed=6, s_begin=2, s_count=3)
    red, sr = comm.render_reduce(cam, w, seed=6, s_begin=2, s_count=3, root=0)
    assert np.array_equal(plain, red)
    for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb"):
        assert sp[k] == sr[k], k
    with pytest.raises(RtError, match="root"):
        comm.reduce_sum(x.data_ptr(), x.numel(), root=1)
    with pytest.raises(RtError, match="rank"):
        Comm(uid, 1, 1)
    comm.close(); w.close()


def test_cli_one_worker_reduces_with_rccl(gpu, tmp_path):
    """`raytracer --gpus 1 --reduce rccl`: a forked worker draws the communicator id, joins the (one-rank) communicator, renders into
    its device framebuffer and sums with ncclReduce — the N-GPU code path — and the image and stats equal the plain run's."""
    exe = os.path.join(os.path.dirname(rt.__file__), "bin", "raytracer")
    outs = []
    for extra in ([], ["--gpus", "1", "--reduce", "rccl"]):
        out = str(tmp_path / ("rccl" if extra else "plain"))
        cmd = [exe, "scenes/dragon_recon/dragon_vrip_res4.ply", "--format", "ply", "--width", "128", "--spp", "4", "--seed", "9", "--output-dir", out] + extra
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=180)
        assert r.returncode == 0, r.stderr
        if extra:
            assert "RCCL ncclReduce" in r.stderr, r.stderr
        outs.append((open(os.path.join(out, "dragon_vrip_res4-bvh.ppm")).read(), list(csv.DictReader(open(os.path.join(out, "stats.csv"))))[0]))
    (p1, r1), (p2, r2) = outs
    assert p1 == p2
    for k in ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb"):
        assert r1[k] == r2[k], k


def test_cli_rccl_refuses_workers_that_share_a_gpu(gpu, tmp_path):
    exe = os.path.join(os.path.dirname(rt.__file__), "bin", "raytracer")
    r = subprocess.run([exe, "scenes/simple_fuzzy_metal.json", "--width", "32", "--spp", "2", "--gpus", "2", "--device", "0", "--reduce", "rccl",
                        "--output-dir", str(tmp_path / "o")], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "one GPU per worker" in (r.stderr + r.stdout)


def test_bench_runs_every_collective_of_the_multi_gpu_path_over_one_rank(gpu):
    """bench.py --dist-at-1 under the driver's launcher (torch.distributed.run, one rank): the `nccl` process group (= RCCL) is
    created on the box's GPU and every collective of the N > 1 path runs over it — the warm-up all-reduce, torch.distributed.reduce
    of the f64 framebuffer after each render, the barriers, the MAX / SUM all-reduces of the timing.  The line must be a normal N = 1
    line with the same rays as a plain run (the sum over one rank is the identity).  N > 1 itself stays unmeasured here."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--gpus", "1", "--steps", "1", "--warmup", "0", "--no-pmc", "--no-cpu-baseline", "--scene", "weekend-final", "--width", "320", "--spp-per-gpu", "8"]
    plain = subprocess.run([sys.executable, "bench.py"] + common, cwd=root, capture_output=True, text=True, timeout=300)
    assert plain.returncode == 0, plain.stderr[-2000:]
    a = json.loads(plain.stdout.strip().splitlines()[-1])
    for reduce in ("torch", "product"):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29541",
                            "bench.py"] + common + ["--dist-at-1", "--reduce", reduce], cwd=root, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        b = json.loads(r.stdout.strip().splitlines()[-1])
        assert b["n_gpus"] == 1 and b["rays_per_step"] == a["rays_per_step"] and b["value"] > 0, (reduce, b)
