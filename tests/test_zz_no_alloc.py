"""pt_render_batch_static (include/ptcore.h): the entry point that allocates nothing.

Checked against pt_render_batch_into, the existing entry point it is a variant of, through the C ABI, bit
for bit (the bar of every image test here), on two calls in a row of which the second has a smaller input;
then the limit and the sharing of the static array between contexts. Collected last (test_zz_*): the
round's end runs pytest -x.
"""
import numpy as np
import pytest

from pathtracer_amd import _ffi, core, scenes
from tests import util

pytestmark = pytest.mark.gpu


def _into(r, torch, w, h, seed, spp, every, exposure=0.0, clear=(0.0, 0.0, 0.0)):
    """The batch loop through pt_configure + pt_render_batch_into (library-allocated tile scratch)."""
    r.configure(w, h, exposure, clear)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        acc = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
        done = 0
        while done < spp:
            cur = min(every, spp - done)
            r.render_batch_into(acc.data_ptr(), seed, done, cur, side.cuda_stream)
            done, seed = done + cur, seed + 1
    side.synchronize()
    return acc.cpu().numpy()


def _static(r, torch, w, h, seed, spp, every, exposure=0.0, clear=(0.0, 0.0, 0.0)):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        acc = torch.zeros(w * h * 3, dtype=torch.float32, device="cuda")
        side.synchronize()
        free0 = torch.cuda.mem_get_info()[0]
        done = 0
        while done < spp:
            cur = min(every, spp - done)
            r.render_batch_static(acc.data_ptr(), w, h, seed, done, cur, side.cuda_stream, exposure, clear)
            done, seed = done + cur, seed + 1
        side.synchronize()
        free1 = torch.cuda.mem_get_info()[0]
    assert free1 >= free0, "device memory was allocated during pt_render_batch_static"
    return acc.cpu().numpy()


def test_static_entry_point_matches_render_batch_into(scene_dir):
    import torch
    big = util.load_bundle(scenes.write_many_lights_standin(scene_dir, rings=40, segments=64, name="ml_static"))
    small = util.load_bundle(scenes.write_cornell_like(scene_dir, name="room_static"))
    r = core.Renderer(0)
    try:
        # first call: 200 x 136 (425 tiles, ragged edges), three launches of 8 samples: static order, then
        # two launches that read the order the launch before them left in the static array
        r.upload(big.scene, big.bvh, big.lst)
        want1 = _into(r, torch, 200, 136, 5, 24, 8, 0.5, (0.1, 0.2, 0.3))
        r.upload(big.scene, big.bvh, big.lst)  # a fresh tree state for the static run (same untimed start)
        got1 = _static(r, torch, 200, 136, 5, 24, 8, 0.5, (0.1, 0.2, 0.3))
        assert np.array_equal(util.bits(got1), util.bits(want1))
        # second call, smaller input: another scene at 72 x 40 (45 tiles). The static array still holds the
        # 425-entry order, costs and RNG states of the first call; none of it may leak into this one.
        r.upload(small.scene, small.bvh, small.lst)
        got2 = _static(r, torch, 72, 40, 9, 20, 8)
        want2 = _into(r, torch, 72, 40, 9, 20, 8)
        assert np.array_equal(util.bits(got2), util.bits(want2))
        assert np.isfinite(got2).all() and got2.max() > 0
        # same scene and size again, right after: now the order left by the previous call counts
        got3 = _static(r, torch, 72, 40, 9, 20, 8)
        assert np.array_equal(util.bits(got3), util.bits(want2))
        # a second context on the same GPU shares the array: its first call must not trust what r left there
        r2 = core.Renderer(0)
        try:
            r2.upload(big.scene, big.bvh, big.lst)
            got4 = _static(r2, torch, 72, 40, 5, 16, 8)
            want4 = _into(r2, torch, 72, 40, 5, 16, 8)
            assert np.array_equal(util.bits(got4), util.bits(want4))
        finally:
            r2.close()
        # pt_configure'd state is not disturbed by static calls of another size
        r.configure(64, 48)
        r.render_batch(3, 0, 8)
        before = r.read_image()
        _static(r, torch, 200, 136, 5, 8, 8)
        r.clear()
        r.render_batch(3, 0, 8)
        assert np.array_equal(util.bits(r.read_image()), util.bits(before))
    finally:
        r.close()


def test_static_entry_point_limits(scene_dir):
    import torch
    b = util.load_bundle(scenes.write_cornell_like(scene_dir, name="room_static2"))
    r = core.Renderer(0)
    try:
        buf = torch.zeros(16, dtype=torch.float32, device="cuda")
        with pytest.raises(core.CoreError, match="pt_scene_upload"):
            r.render_batch_static(buf.data_ptr(), 2, 2, 1, 0, 1)
        r.upload(b.scene, b.bvh, b.lst)
        # 2048 x 2056 = 256 x 257 tiles = 65792 > 65536: refused before anything is launched
        with pytest.raises(core.CoreError, match="static scratch holds 65536"):
            r.render_batch_static(buf.data_ptr(), 2048, 2056, 1, 0, 1)
        with pytest.raises(core.CoreError):
            r.render_batch_static(buf.data_ptr(), 0, 8, 1, 0, 1)
        with pytest.raises(core.CoreError):
            r.render_batch_static(buf.data_ptr(), 2, 2, 1, 0, 0)
        with pytest.raises(core.CoreError):
            r.render_batch_static(0, 2, 2, 1, 0, 1)
        r.render_batch_static(buf.data_ptr(), 2, 2, 1, 0, 1)  # 12 floats
        r.synchronize()
    finally:
        r.close()
