"""N = 2 with the REAL kernel: two ranks (torch.distributed.run, one process each, both on the box's one GPU, so
the collective is gloo's - RCCL refuses two ranks on one device) render their batches with render_sharded and
combine them with the single sum-reduce; the result equals the one-rank batch loop up to fp32 summation order.
(The RCCL call itself runs under test_sharded_render_through_rccl_one_rank and test_group_render_one_device.)

Collected first on purpose (file name): the ranks are child processes, and this pool forbids starting programs
from a process that has already initialised the GPU.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from pathtracer_amd import _ffi, core

pytestmark = pytest.mark.gpu


def test_two_ranks_render_and_combine_on_the_gpu(tmp_path):
    if core.CONTEXTS_CREATED:
        pytest.skip("GPU already initialised in this process; run this file first or alone")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path)
    env = dict(os.environ, OMP_NUM_THREADS="4")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(_ffi.REPO_ROOT, "tests", "sharded_worker.py"), out],
                         capture_output=True, text=True, timeout=170, env=env)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert open(os.path.join(out, "rank0.txt")).read().split() == ["2", "14"]  # batches 0, 2, 4, 6: 4 + 4 + 4 + 2 samples
    assert open(os.path.join(out, "rank1.txt")).read().split() == ["2", "12"]  # batches 1, 3, 5
    combined, single = np.load(os.path.join(out, "combined.npy")), np.load(os.path.join(out, "single.npy"))
    assert combined.shape == single.shape == (136, 200, 3) and np.isfinite(single).all() and np.isfinite(combined).all()
    err = np.abs(combined.astype(np.float64) - single).max()
    scale = np.abs(single).max()
    assert err <= 2e-6 * scale, (err, scale)  # same samples; flat weighted sum against the running mean
    assert np.abs(combined - single).mean() / np.abs(single).mean() <= 1e-6
    assert not np.array_equal(combined, np.zeros_like(combined)) and single.mean() > 0.1


def test_bench_loop_two_ranks_equals_one_rank(tmp_path):
    """bench.py's OWN loop (plan_job -> plan_rank -> render_local -> combine) with two ranks on the box's one GPU
    (gloo; the driver's N > 1 runs use RCCL) against the same command with one rank: the same fixed job (4 batches of
    5 spp: two whole rounds, so the batches and their seeds are the same at both N), the same image to fp32 summation
    order, value = the whole job's samples / time. (Batches that do NOT fill a round are cut into N parts, which
    changes their seeds: then the image is another estimate of the same picture; tests/test_distributed_cpu.py
    checks those plans.)"""
    import json
    if core.CONTEXTS_CREATED:
        pytest.skip("GPU already initialised in this process; run this file first or alone")
    env = dict(os.environ, OMP_NUM_THREADS="4")
    bench = os.path.join(_ffi.REPO_ROOT, "bench.py")
    common = ["--config", "c1", "--res", "160", "--spp", "5", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-tolerance"]
    lines = {}
    for n in (1, 2):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        img = str(tmp_path / f"bench{n}.npy")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
               "--master-port", str(port), bench, "--gpus", str(n), "--backend", "gloo", "--one-gpu", "--dump-image", img] + common
        res = subprocess.run(cmd, capture_output=True, text=True, timeout=170, env=env)
        assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
        lines[n] = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    one, two = np.load(str(tmp_path / "bench1.npy")), np.load(str(tmp_path / "bench2.npy"))
    assert one.shape == two.shape == (160, 160, 3) and np.isfinite(one).all() and one.mean() > 0
    assert np.abs(two.astype(np.float64) - one).max() <= 2e-6 * np.abs(one).max()
    for n in (1, 2):
        ln = lines[n]
        assert ln["scaling"] == "strong" and ln["n_gpus"] == n and ln["steps"] == 4
        assert ln["config"]["spp_total"] == 20 and ln["config"]["spp_per_batch"] == 5 and ln["config"]["batch_sizes"] == [5, 5, 5, 5]
        assert abs(ln["value"] - 160 * 160 * 20 / ln["elapsed_s"] / 1e6) <= 1e-2 * ln["value"]
    assert lines[1]["config"]["batches_on_rank0"] == 4 and lines[2]["config"]["batches_on_rank0"] == 2
    assert lines[1]["config"]["spp_on_rank0"] == 20 and lines[2]["config"]["spp_on_rank0"] == 10
