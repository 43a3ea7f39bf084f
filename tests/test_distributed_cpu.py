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


def test_bench_job_plan_is_the_same_job_at_every_rank_count():
    """bench.py's strong-scaling job: K batches of the configuration's spp in total, whatever N; a batch is split
    only when there are fewer batches than ranks (c2 at N = 8: 8 x 25 spp), and every rank's
    share follows plan_batches (batch b on rank b mod N, seed 42 + b)."""
    import bench
    for steps, spp, world, want in ((4, 50, 1, (50, 4)), (4, 50, 2, (50, 4)), (4, 50, 4, (50, 4)), (4, 50, 8, (25, 8)),
                                    (8, 128, 8, (128, 8)), (20, 50, 8, (50, 20)), (1, 16, 2, (8, 2)), (3, 50, 8, (16, 10))):
        spp_batch, n = bench.plan_job(steps, spp, world, weak=False)
        assert (spp_batch, n) == want, (steps, spp, world, spp_batch, n)
        total = steps * spp
        plans = [render.plan_batches(total, spp_batch, r, world) for r in range(world)]
        assert sorted(b for p in plans for b, _, _ in p) == list(range(n))
        assert sum(c for p in plans for _, _, c in p) == total          # the same job at every N
        assert all(len(p) >= 1 for p in plans) or n < world
        sizes = [len(p) for p in plans]
        assert max(sizes) - min(sizes) <= 1                              # ragged splits differ by one batch at most
    assert bench.plan_job(4, 50, 8, weak=True) == (50, 32)               # --weak: K batches per rank


def test_group_plan_is_the_reference_loop_sharded_by_batch():
    """pt_group_plan (what pt_group_render follows; pure host arithmetic, no GPU): batch b on rank b mod N with
    seed + b, continuing that rank's running mean; weights are the ranks' shares and sum to 1; ranks without a
    batch weigh 0; the ragged last batch is min(every, remaining) as main.cu:102."""
    from pathtracer_amd import core
    for samples, every in ((200, 50), (26, 4), (1024, 128), (7, 50), (1000, 50), (3, 1)):
        for n in range(1, 9):
            batches, weights = core.group_plan(42, samples, every, n)
            assert [b[0] for b in batches] == list(range(len(batches))) and len(batches) == -(-samples // every)
            assert all(rank == i % n and seed == 42 + i for i, rank, seed, _, _ in batches)
            assert [cur for *_, cur in batches] == [min(every, samples - i * every) for i in range(len(batches))]
            local = [0] * n
            for _, rank, _, prev, cur in batches:
                assert prev == local[rank]  # continues that rank's running mean
                local[rank] += cur
            assert sum(local) == samples
            assert abs(sum(np.float64(w) for w in weights) - 1.0) <= n * 2.0 ** -24
            for r in range(n):
                assert weights[r] == np.float32(local[r] / samples)
                assert (weights[r] == 0.0) == (local[r] == 0)
            # the same partition as the torch.distributed path
            for r in range(n):
                assert [(i, i, cur) for i, rank, _, _, cur in batches if rank == r] == render.plan_batches(samples, every, r, n)
    import pytest
    for bad in ((0, 50, 2), (200, 0, 2), (200, 50, 0), (-1, 1, 1)):
        with pytest.raises(core.CoreError):
            core.group_plan(42, *bad)
