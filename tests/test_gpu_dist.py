"""The N>1 product path on hardware: two ranks on the box's one GPU, gloo between them, each calling
rt_render_device for its spp_range and then reduce_framebuffer — the calls bench.py makes per step — against one process
rendering all samples.  "1/2/4/8-GPU runs must produce the same image up to f64 summation order"."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from raytracer_amd import Camera

from .util import load_world

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("scene,width,spp,world", [("weekend-final", 96, 6, 2), ("dragon_res4", 64, 5, 3)])
def test_spp_split_over_ranks_equals_single_process(gpu, tmp_path, scene, width, spp, world):
    w = load_world(scene); w.algorithm = "bvh"; w.init().upload()
    cam = Camera(w.filename, image_width=width, samples_per_pixel=spp)
    full, fst = cam.render(w, seed=13)
    w.close()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "reduced")
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out, w.filename, str(width), str(spp), "13"],
                                      env=env, cwd=ROOT))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out + ".npy"); cnt = json.load(open(out + ".json"))
    for k, v in cnt.items():
        assert v == fst[k], f"{k}: ranks summed {v}, single process {fst[k]}"
    assert np.max(np.abs(got - full) / np.maximum(np.abs(full), 1e-300)) <= 1e-13
