"""bin/pathtracer end to end on the GPU: the reference CLI's flags, batch loop, log lines and
render.png (src/main.cu:36-156), checked against the oracle + the host tonemapper.

Collected before test_gpu_parity.py on purpose: the CLI is started as a child process, and this
pool forbids starting programs from a process that already initialised the GPU, so the test
skips itself if a Renderer was created earlier in this process.
"""
import os
import subprocess

import numpy as np
import pytest

from oracle import binding as ob
from pathtracer_amd import _ffi, core, host, scenes
from tests import util

pytestmark = pytest.mark.gpu

BIN = os.path.join(_ffi.REPO_ROOT, "bin", "pathtracer")


def test_cli_renders_the_oracle_image(scene_dir, tmp_path):
    if core.CONTEXTS_CREATED:
        pytest.skip("GPU already initialised in this process; run this file first or alone")
    assert os.path.exists(BIN), "bin/pathtracer is missing: run __graft_entry__.build()"
    gltf = scenes.write_cornell_like(scene_dir, name="cli_room", detail=6)
    out = str(tmp_path / "out")
    w, h, spp, every, seed = 72, 56, 10, 4, 17
    argv = [BIN, "--path-gltf", gltf, "--dir-output", out, "--output-resolution-x", str(w), "--output-resolution-y", str(h),
            "--sampling-samples", str(spp), "--sampling-samples-every-update", str(every), "--sampling-seed", str(seed),
            "--output-exposure", "0.5", "--world-clear-color", "0 0 0", "--world-clear-color-texture", "",
            "--logger-log-level", "3", "--logger-log-stdout", "0",
            "--default-camera-position", "0 0 5", "--default-camera-target", "0 0 0", "--default-camera-updir", "0 1 0",
            "--default-camera-yfov", "0.7"]
    res = subprocess.run(argv, capture_output=True, text=True, timeout=110)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    log = open(os.path.join(out, "log.txt")).read()
    assert "Rendering 10 samples in batches of 4, img size (72, 56)" in log  # main.cu:93-94
    assert log.count("Rendered ") == 3 and "Rendered 10 out of 10 S/px" in log
    png = host.load_png(os.path.join(out, "render.png"))
    assert png.shape == (h, w, 4)

    b = util.load_bundle(gltf)
    o = ob.Oracle(b.scene, b.bvh, b.lst, w, h, 0.5, (0.0, 0.0, 0.0))
    expect = ob.tonemap_rgb8(o.render(seed, spp, every))  # the ORACLE's tonemapper (rows flipped as image.cu:94 does)
    assert np.array_equal(png[..., :3], expect)

    # --gpus 1: the multi-GPU path of the CLI (pt_group_render: batches sharded over the listed devices, one RCCL
    # sum-reduce to rank 0, one PNG at the end). With one device the picture is the same, byte for byte.
    out2 = str(tmp_path / "out_group")
    res = subprocess.run([a if a != out else out2 for a in argv] + ["--gpus", "1"], capture_output=True, text=True, timeout=110)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    log2 = open(os.path.join(out2, "log.txt")).read()
    assert "on 1 GPUs" in log2 and "Rendered 10 out of 10 S/px" in log2
    assert np.array_equal(host.load_png(os.path.join(out2, "render.png")), png)

    # --gpus N beyond the devices of the node: refused with a message that names both numbers, before any upload
    out3 = str(tmp_path / "out_too_many")
    res = subprocess.run([a if a != out else out3 for a in argv] + ["--gpus", "64"], capture_output=True, text=True, timeout=110)
    assert res.returncode != 0 and "--gpus 64" in res.stderr and "HIP device" in res.stderr and "0..63" in res.stderr
    assert not os.path.exists(os.path.join(out3, "render.png"))
