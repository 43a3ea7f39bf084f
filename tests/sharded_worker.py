"""One rank of tests/test_a_gpu_two_ranks.py (launched by torch.distributed.run, backend gloo, every rank on GPU 0):
render_sharded with the real kernel on this rank's batches, combined across ranks; rank 0 also renders the whole
image alone and writes both to the directory named by argv[1]."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import core, render  # noqa: E402
from tests import util  # noqa: E402

W, H, SEED, SAMPLES, EVERY = 200, 136, 42, 26, 4  # 7 batches: 4 on rank 0, 3 on rank 1 of two; ragged last batch


def main():
    out = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    b = util.golden_scene("glass_mirror", env=True)  # emitter + point light + environment
    r = core.Renderer(0)
    r.upload(b.scene, b.bvh, b.lst)
    r.configure(W, H, 0.5, (1.0, 1.0, 1.0))
    img = render.render_sharded(r, W, H, SEED, SAMPLES, EVERY)
    torch.cuda.synchronize()
    local = sum(c for _, _, c in render.plan_batches(SAMPLES, EVERY, rank, world))
    if rank == 0:
        np.save(os.path.join(out, "combined.npy"), img.cpu().numpy())
        np.save(os.path.join(out, "single.npy"), r.render(SEED, SAMPLES, EVERY))
    with open(os.path.join(out, f"rank{rank}.txt"), "w") as f:
        f.write(f"{world} {local}\n")
    dist.barrier()
    r.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
