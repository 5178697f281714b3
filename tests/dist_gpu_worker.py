"""One rank of tests/test_gpu_dist.py: the N>1 product path on one GPU — rt_render_device for this rank's samples, then the
framebuffer reduce bench.py uses (raytracer_amd.dist.reduce_framebuffer), over gloo.  Rank 0 writes the reduced sum and the
summed counters to argv[1].  Not collected by pytest (no test_ prefix)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)


def main():
    out, scene, width, spp, seed = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)                                   # every rank on the box's one GPU
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracer_amd as rt
    from raytracer_amd import Camera, HittableList
    from raytracer_amd.dist import reduce_framebuffer, spp_range
    rt.init(0)
    w = HittableList.from_ply(scene) if scene.endswith(".ply") else HittableList.from_json(scene)
    w.algorithm = "bvh"
    w.init().upload()
    cam = Camera(scene, image_width=width, samples_per_pixel=spp)
    fb = torch.zeros((cam.image_height, cam.image_width, 3), dtype=torch.float64, device="cuda")
    b, c = spp_range(rank, world, spp)
    _, st = cam.render(w, seed=seed, s_begin=b, s_count=c, d_out=fb.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    reduce_framebuffer(fb, dst=0)
    keys = ("primary_rays", "scatter_rays", "intersection_checks", "traversal_steps", "overlapping_aabb")
    cnt = torch.tensor([float(st[k]) for k in keys], dtype=torch.float64)
    dist.all_reduce(cnt)
    if rank == 0:
        np.save(out + ".npy", fb.cpu().numpy())
        json.dump({k: int(v) for k, v in zip(keys, cnt.tolist())}, open(out + ".json", "w"))
    dist.barrier()
    dist.destroy_process_group()
    w.close()


if __name__ == "__main__":
    main()
