"""N > 1 path on CPU: batch sharding + the single sum-reduce, world_size 2, gloo backend.

The per-rank renderer is the oracle here (no GPU in this tier); what is under test is
pathtracer_amd.render: which batches a rank takes, how the per-rank running means are weighted and
combined, and that the result equals the single-process batch loop up to fp32 summation order.
"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pathtracer_amd import render
from tests import util

W, H, SEED, SAMPLES, EVERY = 40, 32, 42, 14, 4  # batches of 4, 4, 4, 2


def test_plan_batches_partitions_every_batch_once():
    for world in (1, 2, 3, 8):
        seen = []
        for rank in range(world):
            plan = render.plan_batches(SAMPLES, EVERY, rank, world)
            assert all(b % world == rank for b, _, _ in plan)
            seen += plan
        seen.sort()
        assert [b for b, _, _ in seen] == list(range(4))
        assert [c for _, _, c in seen] == [4, 4, 4, 2]
        assert [s for _, s, _ in seen] == [0, 1, 2, 3]  # seed + b, as main.cu:121
    assert render.plan_batches(0, 4) == []
    assert render.plan_batches(3, 50) == [(0, 0, 3)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path):
    from oracle import binding as ob
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = util.golden_cornell()
    o = ob.Oracle(b.scene, b.bvh, b.lst, W, H)
    img = np.zeros((H, W, 3), np.float32)
    prev = 0
    for _, seed_offset, cur in render.plan_batches(SAMPLES, EVERY, rank, world):
        o.render_batch(SEED + seed_offset, prev, cur, img, threads=1)
        prev += cur
    acc = torch.from_numpy(img.reshape(-1))
    render.combine(acc, prev, SAMPLES)
    if rank == 0:
        np.save(out_path, acc.numpy().reshape(H, W, 3))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_reduce_matches_single_process(tmp_path):
    from oracle import binding as ob
    out = str(tmp_path / "combined.npy")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    combined = np.load(out)
    b = util.golden_cornell()
    single = ob.Oracle(b.scene, b.bvh, b.lst, W, H).render(SEED, SAMPLES, EVERY, threads=2)
    assert np.isfinite(combined).all()
    # same samples, different fp32 summation order (flat weighted sum vs running mean)
    err = np.abs(combined.astype(np.float64) - single).max()
    scale = np.abs(single).max()
    assert err <= 2e-6 * scale, (err, scale)
    rel_mean = np.abs(combined - single).mean() / np.abs(single).mean()
    assert rel_mean <= 1e-6
