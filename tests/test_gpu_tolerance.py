"""Tolerance arithmetic (pt_set_arithmetic(PT_ARITHMETIC_FAST)) graded by north_star's three checks, NOT by bit-identity:

  1. the integer RNG stream is bit-exact;
  2. primary-ray hit triangle ids equal the oracle's apart from a STATED count of epsilon edge ties;
  3. the image is within a STATED per-pixel RMSE / mean relative error of the oracle's.

The exact variant stays the gate (every other GPU test); this file states what the fast variant gives up.
Measured on the box (round 3, profiles/r03_tolerance.log):
  * ids, jittered rays (what a render traces): cornell 256^2 1 of 65 536, dragon 1024^2 16 of 1 048 576, glass_mirror
    1024^2 0, a grid of 180 000 tiny triangles 22; without jitter the pixel grid meets the axis-aligned meshes EXACTLY on
    their edges, where exact arithmetic gives u == 0 in both neighbours and a fused one +-1e-9: 62 / 18 / 131 / 727 (224 on the grid in an earlier build: which products
    the compiler fuses changes with the surrounding code, and these rays sit exactly on the edges).
  * image: a path only changes when a differently rounded number flips a DECISION (rejection test, Russian roulette,
    an edge), which a 1e-7 relative change rarely does; otherwise the pixel differs in its last bits. C1 (cornell
    256^2 x 16 spp): per-pixel RMSE 7e-8 = 1e-6 of the image mean, mean relative error 1e-8, against an RMSE of 0.094
    between two oracle seeds. C2 (many_lights stand-in 1024^2 x 200 spp): RMSE 3.1e-4 = 0.003 of the image mean (a
    handful of pixels whose paths did flip), mean relative error 1.3e-6, against 0.054 between two seeds.
The bounds asserted below leave room for other flips than the ones this run met.
"""
import os

import numpy as np
import pytest

from oracle import binding as ob
from pathtracer_amd import core, scenes
from tests import util

pytestmark = pytest.mark.gpu

THREADS = 16


def _renderer(b, w, h, arithmetic, clear=(0.0, 0.0, 0.0), exposure=0.0):
    r = core.Renderer(0)
    r.set_arithmetic(arithmetic)
    r.upload(b.scene, b.bvh, b.lst)
    r.configure(w, h, exposure, clear)
    assert r.arithmetic() == arithmetic
    return r


def _compare(a, ref):
    """(fraction of floats identical, RMSE, RMSE / mean(ref), |mean(a) - mean(ref)| / mean(ref)) over the floats finite in both."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    ok = np.isfinite(a) & np.isfinite(ref)
    same = float((a[ok] == ref[ok]).mean())
    rmse = float(np.sqrt(np.mean((a[ok] - ref[ok]) ** 2)))
    m = float(ref[ok].mean())
    return same, rmse, rmse / m, abs(float(a[ok].mean()) - m) / m


def test_fast_arithmetic_keeps_the_integer_rng_stream(cornell_bundle):
    r = _renderer(cornell_bundle, 64, 64, "fast")
    try:
        for seed, sub in ((42, 0), (42, 1), (43, 64 * 64 - 1), (-3, 12345), (2 ** 31 - 1, 2 ** 24 + 5)):
            assert np.array_equal(r.rng_stream(seed, sub, 64), ob.rng_stream(seed, sub, 64))
    finally:
        r.close()


@pytest.fixture(scope="module")
def cornell_bundle():
    return util.golden_scene("cornell")


# (label, bundle maker, size, clear, allowed id mismatches without jitter, with jitter)
def _hit_cases(scene_dir):
    return [
        ("C1 cornell.gltf 256x256", lambda: util.golden_scene("cornell"), 256, (0.0, 0.0, 0.0)),
        ("C3 dragon.gltf 1024x1024", lambda: util.golden_scene("dragon"), 1024, (0.0, 0.0, 0.0)),
        ("C4 glass_mirror.gltf + env 1024x1024", lambda: util.golden_scene("glass_mirror", env=True), 1024, (1.0, 1.0, 1.0)),
        ("grid of 180,000 tiny triangles 1024x1024", lambda: util.load_bundle(scenes.write_synthetic_grid(scene_dir, name="tol_grid", side=300)), 1024,
         (0.0, 0.0, 0.0)),
    ]


def test_fast_primary_hit_ids_differ_from_the_oracle_only_in_edge_ties(scene_dir):
    """Stated counts (printed; measured values in the module docstring): a ray whose id differs hits the neighbouring
    triangle across an edge at the same distance to 1e-4 relative, or grazes a silhouette (a depth edge of the mesh
    against itself, or hit on one side and miss on the other)."""
    for label, make, res, clear in _hit_cases(scene_dir):
        b = make()
        o = ob.Oracle(b.scene, b.bvh, b.lst, res, res, 0.0, clear)
        for tree in (4, 2):
            r = _renderer(b, res, res, "fast", clear)
            try:
                for jitter in (False, True):
                    ids, dist = r.primary_hits(42, jitter=jitter, tree=tree)
                    oi, od = o.primary_hits(42, jitter=jitter)
                    bad = ids != oi
                    n = int(bad.sum())
                    both = bad & (ids >= 0) & (oi >= 0)
                    rel = np.abs(dist[both] - od[both]) / od[both] if both.any() else np.zeros(0)
                    edge = int((rel <= 1e-4).sum())
                    silhouette = n - int(both.sum())  # hit on one side, miss (or another surface) on the other
                    far = int((rel > 1e-4).sum())      # another surface behind an edge: a silhouette of the mesh against itself
                    print(f"{label} tree {tree} jitter {int(jitter)}: {n} of {ids.size} ids differ "
                          f"({edge} edge ties at equal distance, {far} across a depth edge, {silhouette} hit/miss)")
                    assert n <= (64 if jitter else ids.size // 500), (label, tree, jitter, n)
                    same = ~bad & (ids >= 0)
                    worst = float(np.max(np.abs(dist[same] - od[same]) / od[same], initial=0.0))
                    print(f"    largest relative distance difference where the ids agree: {worst:.3g}")
                    assert worst <= 5e-4  # measured 4.5e-5 (grazing triangles of the dragon)
            finally:
                r.close()


def test_fast_image_error_c1_cornell_16spp(cornell_bundle):
    """BASELINE config 1 (cornell.gltf 256x256, 16 spp, seed 42): fast variant against the oracle."""
    b = cornell_bundle
    o = ob.Oracle(b.scene, b.bvh, b.lst, 256, 256)
    cpu = o.render(42, 16, 16, threads=THREADS)
    other_seed = o.render(43, 16, 16, threads=THREADS)
    r = _renderer(b, 256, 256, "fast")
    try:
        gpu = r.render(42, 16, 16)
    finally:
        r.close()
    same, rmse, rel_rmse, rel_mean = _compare(gpu, cpu)
    _, noise, rel_noise, noise_mean = _compare(other_seed, cpu)
    print(f"C1 fast vs oracle: {same * 100:.2f} % of floats identical, per-pixel RMSE {rmse:.4g} ({rel_rmse:.3f} of the image mean), "
          f"mean relative error {rel_mean:.3g}; two oracle seeds: RMSE {noise:.4g} ({rel_noise:.3f}), mean {noise_mean:.3g}")
    assert rmse <= 0.05 * noise  # measured 7e-7 of it: no path flipped a decision, every pixel agrees to its last bits
    assert rel_mean <= 1e-4


def test_fast_image_error_c2_many_lights_1024_200spp(scene_dir):
    """BASELINE config 2 (many_lights stand-in 1024x1024, 200 spp in batches of 50, seed 42): fast variant against the
    exact one, which is bit-identical to the oracle at this size (tests/test_gpu_full_size.py)."""
    b = util.load_bundle(scenes.write_many_lights_standin(scene_dir, name="ml_tol"))
    imgs, ms = {}, {}
    for arithmetic, seed in (("exact", 42), ("fast", 42), ("exact2", 1042)):
        r = _renderer(b, 1024, 1024, arithmetic.rstrip("2"))
        try:
            imgs[arithmetic] = r.render(seed, 200, 50)
            ms[arithmetic] = r.last_kernel_ms()
        finally:
            r.close()
    same, rmse, rel_rmse, rel_mean = _compare(imgs["fast"], imgs["exact"])
    _, noise, rel_noise, noise_mean = _compare(imgs["exact2"], imgs["exact"])
    print(f"C2 fast vs exact (= oracle): {same * 100:.2f} % of floats identical, per-pixel RMSE {rmse:.4g} ({rel_rmse:.3f} of the image "
          f"mean), mean relative error {rel_mean:.3g}; two seeds of the exact kernel: RMSE {noise:.4g} ({rel_noise:.3f}), mean "
          f"{noise_mean:.3g}; ms per 50-spp launch: exact {ms['exact']:.2f}, fast {ms['fast']:.2f}")
    assert rmse <= 0.1 * noise  # measured 0.006 of it
    assert rel_mean <= 1e-4
    assert ms["fast"] < ms["exact"]


def test_switching_arithmetic_back_restores_bit_identity(cornell_bundle):
    b = cornell_bundle
    cpu = ob.Oracle(b.scene, b.bvh, b.lst, 96, 80).render(7, 6, 3, threads=THREADS)
    r = _renderer(b, 96, 80, "exact")
    try:
        first = r.render(7, 6, 3)
        r.set_arithmetic("fast")
        r.render(7, 6, 3)
        r.set_arithmetic("exact")
        again = r.render(7, 6, 3)
    finally:
        r.close()
    assert util.count_different(first, cpu) == 0 and util.count_different(again, cpu) == 0


def test_fast_arithmetic_through_every_entry_point_agrees_with_itself(cornell_bundle):
    """The tolerance variant behind each way of launching it: pt_render_batch, pt_render_batch_into on a caller stream,
    pt_render_batch_static and a pt_group of one device give the same floats (its arithmetic is deterministic: tile
    order, tree form and entry point change nothing), and an unknown arithmetic is refused."""
    import torch
    b = cornell_bundle
    w, h, seed, spp, every = 88, 72, 11, 9, 4
    r = _renderer(b, w, h, "fast")
    try:
        want = r.render(seed, spp, every)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            acc = torch.zeros(h * w * 3, dtype=torch.float32, device="cuda")
            acc2 = torch.zeros_like(acc)
            done, s = 0, seed
            while done < spp:
                cur = min(every, spp - done)
                r.render_batch_into(acc.data_ptr(), s, done, cur, side.cuda_stream)
                r.render_batch_static(acc2.data_ptr(), w, h, s, done, cur, side.cuda_stream)
                done, s = done + cur, s + 1
            side.synchronize()
        assert np.array_equal(util.bits(acc.cpu().numpy().reshape(h, w, 3)), util.bits(want))
        assert np.array_equal(util.bits(acc2.cpu().numpy().reshape(h, w, 3)), util.bits(want))
        for tree in ("binary", "wide"):  # the other form of the tree: same floats (no tie bookkeeping, but the same candidates)
            os.environ["PT_TREE"] = tree
            try:
                r2 = _renderer(b, w, h, "fast")
                try:
                    other = r2.render(seed, spp, every)
                finally:
                    r2.close()
            finally:
                del os.environ["PT_TREE"]
            same = float((util.bits(other) == util.bits(want)).mean())
            print(f"fast arithmetic, PT_TREE={tree}: {same * 100:.2f} % of floats identical to the timed choice")
            assert same > 0.9  # a walk's winner can differ between the trees only in an exact tie
        with pytest.raises(core.CoreError):
            core._check(r._lib.pt_set_arithmetic(r._ctx, 7))
        assert r.arithmetic() == "fast"
    finally:
        r.close()
    g = core.Group([0])
    try:
        for rr in g.ranks:
            rr.set_arithmetic("fast")
        g.upload(b.scene, b.bvh, b.lst)
        g.configure(w, h)
        assert np.array_equal(util.bits(g.render(seed, spp, every)), util.bits(want))
    finally:
        g.close()
