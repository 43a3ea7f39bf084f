"""GPU parity: the HIP kernel (through the C ABI) against the CPU oracle on the same inputs.

Bars (BASELINE.json north_star):
  * integer RNG stream: bit-exact;
  * primary-ray hit triangle ids: bit-exact (0 mismatches; distances compared as bits too);
  * image: the oracle and the kernel round every float operation identically
    (-ffp-contract=off, shared deterministic transcendentals), so the linear radiance image
    is compared for exact equality, which is stronger than the RMSE bar; an RMSE/relative
    bound is asserted as well so a future relaxation has a stated tolerance to fall back to.
"""
import numpy as np
import pytest

from oracle import binding as ob
from pathtracer_amd import _ffi, core, host, scenes
from tests import util

pytestmark = pytest.mark.gpu

RMSE_TOL = 1e-6       # per-pixel RMSE of linear radiance, stated tolerance of the image check
MEAN_REL_TOL = 1e-6   # mean relative error


@pytest.fixture(scope="module")
def renderer():
    r = core.Renderer(0)
    yield r
    r.close()


def _setup(renderer, bundle, w, h, exposure=0.0, clear=(0.0, 0.0, 0.0)):
    renderer.upload(bundle.scene, bundle.bvh, bundle.lst)
    renderer.configure(w, h, exposure, clear)
    return ob.Oracle(bundle.scene, bundle.bvh, bundle.lst, w, h, exposure, clear)


def _assert_images_match(gpu, cpu, allow_nan=False):
    """allow_nan: the reference's own scenes have roughness-0 materials, where GGX is 0/0 when the half vector
    falls on the normal (brdf.cu:13-14): such a pixel must be NaN on BOTH sides."""
    if allow_nan:
        both = np.isnan(cpu) & np.isnan(gpu)
        assert both.mean() < 1e-3
        assert util.count_different(gpu, cpu) == 0, f"{util.count_different(gpu, cpu)} of {gpu.size} floats differ"
        gpu, cpu = np.where(both, 0, gpu), np.where(both, 0, cpu)
    assert np.isfinite(cpu).all() and np.isfinite(gpu).all()
    diff = gpu.astype(np.float64) - cpu.astype(np.float64)
    rmse = float(np.sqrt((diff ** 2).mean()))
    rel = float(np.abs(diff).mean() / max(np.abs(cpu).mean(), 1e-30))
    assert rmse <= RMSE_TOL and rel <= MEAN_REL_TOL, (rmse, rel)
    mismatched = int((gpu != cpu).sum())
    assert mismatched == 0, f"{mismatched} of {gpu.size} floats differ (rmse {rmse:.3e})"


def test_rng_stream_bit_exact(renderer):
    for seed in (42, 43, 0, -1, 2 ** 40 + 7):
        for sub in (0, 1, 2, 3, 63, 64, 255 * 256 + 255, 1024 * 1024 - 1, 2 ** 31 + 12345, 2 ** 32 - 1):
            g = renderer.rng_stream(seed, sub, 64)
            c = ob.rng_stream(seed, sub, 64)
            assert np.array_equal(g, c), (seed, sub)


@pytest.fixture(scope="module")
def cornell_like(scene_dir):
    return util.load_bundle(scenes.write_cornell_like(scene_dir))


def _assert_primary_hits_match(renderer, o, seed, jitter=True):
    """north_star check 2: primary-ray hit triangle ids bit-exact, through the PRODUCT walker (sliced
    speculative walk + sequential start-over, pt_primary_hits) on both forms of the tree. Returns the
    stated count of epsilon ties per tree: rays that found two candidates within rounding of each other and
    were walked again in the reference's order (their ids are compared like all others)."""
    ci, cd = o.primary_hits(seed, jitter)
    ties = {}
    for tree in (2, 4):
        gi, gd, ties[tree] = renderer.primary_hits(seed, jitter, tree=tree, with_ties=True)
        assert int((gi != ci).sum()) == 0, (tree, int((gi != ci).sum()))
        assert np.array_equal(util.bits(gd), util.bits(cd)), tree
    return ci, ties


def test_primary_hits_bit_exact(renderer, cornell_like, scene_dir):
    w = h = 160
    o = _setup(renderer, cornell_like, w, h)
    for jitter in (False, True):
        ci, ties = _assert_primary_hits_match(renderer, o, 42, jitter)
        assert (ci >= 0).mean() > 0.9  # the room fills the frame
        print(f"cornell-like {w}x{h} jitter={jitter}: 0 mismatches, epsilon ties walked again: {ties}")
    # the height field of tiny triangles (the shape of BASELINE config 5): here rays DO meet candidate pairs
    # within rounding of each other, and the start-over is what keeps the ids equal
    b = util.load_bundle(scenes.write_synthetic_grid(scene_dir, name="grid_hits", side=300, emitters=4))
    o = _setup(renderer, b, 320, 240)
    ci, ties = _assert_primary_hits_match(renderer, o, 42, True)
    print(f"grid 300x300 quads 320x240: 0 mismatches, epsilon ties walked again: {ties}")
    assert ties[2] > 0 and ties[4] > 0


def test_cornell_like_image_exact(renderer, cornell_like):
    w = h = 96
    o = _setup(renderer, cornell_like, w, h)
    renderer.clear()
    renderer.render_batch(42, 0, 8)
    cpu, _ = o.render_batch(42, 0, 8)
    _assert_images_match(renderer.read_image(), cpu)


def test_read_image_into_page_locked_memory(renderer, cornell_like):
    """pt_host_alloc / pt_host_free (what bin/pathtracer reads its batches into): the same floats as a read into
    pageable memory, and a null pointer frees as nothing."""
    import ctypes as C
    w, h = 72, 40
    _setup(renderer, cornell_like, w, h)
    renderer.clear()
    renderer.render_batch(5, 0, 3)
    want = renderer.read_image()
    lib = renderer._lib
    p = lib.pt_host_alloc(w * h * 3 * 4)
    assert p
    try:
        assert lib.pt_read_image(renderer._ctx, p) == 0
        got = np.frombuffer((C.c_float * (w * h * 3)).from_address(p), np.float32).reshape(h, w, 3).copy()
    finally:
        lib.pt_host_free(p)
    lib.pt_host_free(None)
    assert np.array_equal(util.bits(got), util.bits(want))


def test_running_mean_over_batches(renderer, cornell_like):
    """main.cu:101-133: seed + 1 per batch, running mean, ragged last batch."""
    w, h = 80, 48  # not multiples of the 16x16 block
    o = _setup(renderer, cornell_like, w, h, exposure=1.5)
    gpu = renderer.render(7, 11, 4)  # batches of 4, 4, 3
    cpu = o.render(7, 11, 4)
    _assert_images_match(gpu, cpu)


def test_tile_reordering_keeps_pixels(renderer, scene_dir):
    """From the second launch on the tiles are dispatched heaviest-first (per-tile cost of the
    previous launch); every pixel must still come out as in the oracle, at a size with more
    tiles than wave slots per CU and a ragged right/top edge."""
    b = util.load_bundle(scenes.write_many_lights_standin(scene_dir, rings=40, segments=64, name="ml_small"))
    w, h = 200, 136
    o = _setup(renderer, b, w, h)
    gpu = renderer.render(5, 8, 2)  # four launches: static order, then three re-ranked ones
    cpu = o.render(5, 8, 2)
    _assert_images_match(gpu, cpu)


def test_golden_cornell_c1(renderer):
    """BASELINE config 1 (cornell.gltf 256x256, 16 spp, seed 42) on the reference's own scene,
    carried to the GPU box as derived arrays."""
    b = util.golden_cornell()
    o = _setup(renderer, b, 256, 256)
    gpu = renderer.render(42, 16, 16)
    cpu = o.render(42, 16, 16)
    _assert_images_match(gpu, cpu)
    _assert_primary_hits_match(renderer, o, 42, True)


def test_golden_dragon_c3_small(renderer):
    """BASELINE config 3's scene, the reference's own dragon.gltf (139 482 faces, no TEXCOORD_0: constant
    tangent (1,0,0), skewed shading basis, non-unit bounce directions; roughness-0 materials), carried to the
    GPU box as derived arrays. Small size here; 1024x1024 in tests/test_gpu_full_size.py."""
    b = util.golden_scene("dragon")
    assert len(b.scene.faces) == 139482 and b.bvh.stats["node_count"] == 168503
    o = _setup(renderer, b, 192, 192)
    gpu = renderer.render(42, 16, 8)
    cpu = o.render(42, 16, 8)
    _assert_images_match(gpu, cpu, allow_nan=True)
    _assert_primary_hits_match(renderer, o, 42, True)


def test_golden_glass_mirror_env_c4_small(renderer):
    """BASELINE config 4: the reference's glass_mirror.gltf (point light, roughness-0 metal and glass) under
    its sunny_rose_garden_1k.hdr environment (decoded texels, point-sampled to 256x128), clear colour 1 1 1."""
    b = util.golden_scene("glass_mirror", env=True)
    assert len(b.scene.faces) == 1932 and len(b.scene.lights) == 1 and len(b.lst.nodes) == 3
    o = _setup(renderer, b, 192, 192, clear=(1.0, 1.0, 1.0))
    gpu = renderer.render(42, 16, 8)
    cpu = o.render(42, 16, 8)
    _assert_images_match(gpu, cpu, allow_nan=True)
    assert np.nanmean(gpu) > 0.2  # the environment lights the scene
    _assert_primary_hits_match(renderer, o, 42, True)


def test_punctual_lights_and_env_map(renderer, scene_dir):
    """Point light NEE, mirrored-repeat bilinear env map on misses, non-zero exposure."""
    hdr = scenes.write_sky_hdr(scene_dir + "/sky.hdr")
    b = util.load_bundle(scenes.write_glass_mirror_standin(scene_dir), clear_texture=hdr)
    o = _setup(renderer, b, 64, 64, exposure=2.0, clear=(1.0, 1.0, 1.0))
    renderer.clear()
    renderer.render_batch(3, 0, 6)
    cpu, _ = o.render_batch(3, 0, 6)
    _assert_images_match(renderer.read_image(), cpu)


def test_directional_light_and_large_bvh(renderer, scene_dir):
    b = util.load_bundle(scenes.write_many_lights_standin(scene_dir, rings=60, segments=96))
    o = _setup(renderer, b, 64, 64)
    renderer.clear()
    renderer.render_batch(42, 0, 4)
    cpu, _ = o.render_batch(42, 0, 4)
    _assert_images_match(renderer.read_image(), cpu)


def test_default_tangents_no_uv(renderer, scene_dir):
    """dragon-like mesh without TEXCOORD_0: skewed shading basis, non-unit bounce directions."""
    b = util.load_bundle(scenes.write_dragon_standin(scene_dir, rings=48, segments=48))
    o = _setup(renderer, b, 64, 48)
    renderer.clear()
    renderer.render_batch(42, 0, 6)
    cpu, _ = o.render_batch(42, 0, 6)
    _assert_images_match(renderer.read_image(), cpu)


def test_sharded_render_single_rank(renderer, cornell_like):
    """render.render_sharded (the torch-side accumulator + combine used for N GPUs) with one rank
    must equal the library-owned accumulator path and the oracle, bit for bit."""
    import torch
    from pathtracer_amd import render
    w, h = 72, 40
    o = _setup(renderer, cornell_like, w, h)
    img = render.render_sharded(renderer, w, h, 9, 10, 4)  # batches of 4, 4, 2
    torch.cuda.synchronize()
    gpu = img.cpu().numpy()
    cpu = o.render(9, 10, 4)
    _assert_images_match(gpu.reshape(cpu.shape), cpu)


def test_sharded_render_through_rccl_one_rank(renderer, cornell_like):
    """the N-GPU path's collective on hardware. A one-rank "nccl" (= RCCL) process group:
    render_sharded launches on its side stream and combine() runs the real all_reduce on the accumulator.
    (The N > 1 arithmetic - plan, weights, sum - is covered by the gloo world-2 test on CPU.)"""
    import socket
    import torch
    import torch.distributed as dist
    from pathtracer_amd import render
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    w, h = 72, 40
    o = _setup(renderer, cornell_like, w, h)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        img = render.render_sharded(renderer, w, h, 9, 10, 4)
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)  # the group is live
        torch.cuda.synchronize()
        assert float(t.sum()) == 4.0
        gpu = img.cpu().numpy()
    finally:
        dist.destroy_process_group()
    _assert_images_match(gpu, o.render(9, 10, 4))


def test_group_render_one_device(cornell_like):
    """pt_group_*: the single-process multi-GPU path (one context per device, batches b = rank mod N, weights,
    ONE ncclReduce to rank 0) with the one device this box has: the image is the oracle's, bit for bit.
    A device listed twice is refused (RCCL needs distinct devices)."""
    w, h = 88, 48
    g = core.Group([0])
    try:
        g.upload(cornell_like.scene, cornell_like.bvh, cornell_like.lst)
        g.configure(w, h, 0.5)
        gpu = g.render(9, 11, 4)  # batches of 4, 4, 3
        gpu2 = g.render(9, 11, 4)  # and again: the accumulators start from zero
    finally:
        g.close()
    cpu = ob.Oracle(cornell_like.scene, cornell_like.bvh, cornell_like.lst, w, h, 0.5).render(9, 11, 4)
    _assert_images_match(gpu, cpu)
    _assert_images_match(gpu2, cpu)
    with pytest.raises(core.CoreError, match="listed twice"):
        core.Group([0, 0])


def test_material_textures_and_alpha_modes(renderer, scene_dir):
    """Base-colour + metallic-roughness textures (mirrored / repeat / clamp wrap, linear and
    nearest filter, uvs outside [0,1]), MASK cutoff and BLEND stochastic pass-through."""
    b = util.load_bundle(scenes.write_textured_standin(scene_dir))
    m = b.scene.materials
    assert (m["baseColorTexture"] >= 0).sum() == 2 and (m["metallicRoughnessTexture"] >= 0).sum() == 1
    o = _setup(renderer, b, 96, 80)
    renderer.clear()
    renderer.render_batch(11, 0, 6)
    cpu, c = o.render_batch(11, 0, 6)
    _assert_images_match(renderer.read_image(), cpu)
    g = renderer.count_batch(11, 6)
    assert g["rays"] == c["rays"] and g["shaded_hits"] == c["shaded_hits"]


class _ArraysBvh:
    def __init__(self, nodes, indices):
        self.nodes, self.indices = nodes, indices


def _chain_bvh(scene):
    """Degenerate but valid tree: node k = (leaf of face k, node k+1). Height = face count."""
    v, f = scene.vertices, scene.faces
    tri = v["position"][f["vertices"]]                      # (F, 3, 3)
    lo, hi = tri.min(axis=1), tri.max(axis=1)
    F = len(f)
    nodes = np.zeros(2 * F - 1, _ffi.BVH_NODE_DTYPE)
    suffix_lo, suffix_hi = lo.copy(), hi.copy()
    for k in range(F - 2, -1, -1):
        suffix_lo[k] = np.minimum(lo[k], suffix_lo[k + 1])
        suffix_hi[k] = np.maximum(hi[k], suffix_hi[k + 1])
    for k in range(F - 1):
        inner, leaf = 2 * k, 2 * k + 1
        nodes[inner]["bmin"], nodes[inner]["bmax"] = suffix_lo[k], suffix_hi[k]
        nodes[inner]["leftChild"] = leaf
        nodes[inner]["rightChild"] = 2 * k + 2
        nodes[leaf]["bmin"], nodes[leaf]["bmax"] = lo[k], hi[k]
        nodes[leaf]["start"], nodes[leaf]["end"] = k, k + 1
    last = 2 * F - 2
    nodes[last]["bmin"], nodes[last]["bmax"] = lo[F - 1], hi[F - 1]
    nodes[last]["start"], nodes[last]["end"] = F - 1, F
    return _ArraysBvh(nodes, np.arange(F, dtype=np.uint32))


def test_tall_tree_uses_the_64_entry_stack(renderer, scene_dir):
    """A 50-level chain handed over in the reference's node format: the upload picks the 64-entry
    stack instantiation (bvh.cu:338-342 allows 64), and caller-built trees are honoured as given."""
    b = util.load_bundle(scenes.write_cornell_like(scene_dir, name="tiny_room", detail=2))
    assert 34 < len(b.scene.faces) <= 64
    b.bvh = _chain_bvh(b.scene)
    o = _setup(renderer, b, 48, 40)
    renderer.clear()
    renderer.render_batch(2, 0, 5)
    cpu, _ = o.render_batch(2, 0, 5)
    _assert_images_match(renderer.read_image(), cpu)
    gi, gd = renderer.primary_hits(2, True)
    ci, cd = o.primary_hits(2, True)
    assert int((gi != ci).sum()) == 0 and np.array_equal(util.bits(gd), util.bits(cd))
    gi, gd = renderer.primary_hits(2, True, tree=2)
    assert int((gi != ci).sum()) == 0 and np.array_equal(util.bits(gd), util.bits(cd))


def test_single_leaf_tree_and_one_sample(renderer, scene_dir):
    """Two triangles: the root is a leaf (bvh.cu:143); one sample per launch, 3 launches."""
    w = scenes.GltfWriter()
    w.doc.setdefault("extensionsUsed", []).append("KHR_materials_emissive_strength")
    emit = w.material("EMIT", (0.2, 0.2, 0.2, 1), emissive=(1, 0.5, 0.25), emissive_strength=2)
    w.node("Plane", mesh=w.mesh("Plane", [scenes.unit_plane(w, emit)]), rotation=(0, 0.7071068, 0, 0.7071068))
    w.node("Camera", camera=w.camera(scenes.ROOM_CAMERA["yfov"]), rotation=scenes.ROOM_CAMERA["rotation"],
           translation=scenes.ROOM_CAMERA["translation"])
    b = util.load_bundle(w.write(scene_dir, "two_tris"))
    assert len(b.scene.faces) == 2 and len(b.bvh.nodes) == 1
    o = _setup(renderer, b, 40, 24, clear=(0.1, 0.2, 0.3))
    gpu = renderer.render(3, 3, 1)
    cpu = o.render(3, 3, 1)
    _assert_images_match(gpu, cpu)
    assert gpu.max() > 0.3  # the emitter is in view


def test_four_wide_tree_and_axis_parallel_rays(scene_dir, monkeypatch):
    """PT_TREE=wide forces the four-wide tree (folded from the reference's at upload). A sun that
    shines straight down the -z axis makes every one of its shadow rays axis-parallel (two infinite
    direction reciprocals): those rays must take the binary tree and the reference's exact box test."""
    w = scenes.GltfWriter()
    grey = w.material("GREY", (0.6, 0.6, 0.6, 1), roughness=0.3)
    w.node("Ground", mesh=w.mesh("Ground", [scenes.unit_plane(w, grey)]), scale=(3, 3, 1), translation=(0, 0, -1))
    p, nrm, uv, idx = scenes.blob(10, 20, bump=0.3, seed=11)
    w.node("Blob", mesh=w.mesh("Blob", [w.primitive(p, nrm, idx, grey, uv)]), scale=(0.5,) * 3, translation=(0.1, -0.2, -0.4))
    w.node("Sun", light=w.light("directional", 4.0))  # no rotation: direction (0, 0, -1)
    w.node("Camera", camera=w.camera(scenes.ROOM_CAMERA["yfov"]), rotation=scenes.ROOM_CAMERA["rotation"],
           translation=scenes.ROOM_CAMERA["translation"])
    b = util.load_bundle(w.write(scene_dir, "axis_sun"))
    assert np.allclose(b.scene.lights["direction"][0], (0, 0, -1))
    for tree, expect in (("wide", 4), ("binary", 2)):
        monkeypatch.setenv("PT_TREE", tree)
        r = core.Renderer(0)
        try:
            o = _setup(r, b, 88, 64, clear=(0.2, 0.2, 0.3))
            gpu = r.render(13, 12, 6)
            assert r.walk_tree() == expect
            cpu = o.render(13, 12, 6)
            _assert_images_match(gpu, cpu)
            assert gpu.mean() > 0.05
        finally:
            r.close()


def test_full_size_stand_in_both_trees(scene_dir, monkeypatch):
    """The bench scene itself (322 804 faces, tree height 28: the four-wide walk needs the 48-entry
    stack instantiation), 256x256, 16 spp in two launches, on each tree."""
    b = util.load_bundle(scenes.write_many_lights_standin(scene_dir, name="ml_full"))
    assert len(b.scene.faces) == 322804
    cpu = None
    for tree in ("wide", "binary"):
        monkeypatch.setenv("PT_TREE", tree)
        r = core.Renderer(0)
        try:
            o = _setup(r, b, 256, 256)
            gpu = r.render(42, 16, 8)
            if cpu is None:
                cpu = o.render(42, 16, 8)
            _assert_images_match(gpu, cpu)
        finally:
            r.close()


def test_coincident_faces_keep_the_reference_tie_rule(cornell_like, monkeypatch):
    """Every face three times, the copies in other colours: the copies are hit at exactly the same distance
    and the reference keeps whichever its traversal tests first (strict `<`, intersect.cu:100). Leaves
    hold two faces, so a triple spans two leaves and the order leaves are found in decides. The binary
    tree finds them in the reference's order; the four-wide one sorts four children by distance, which need
    not be that order. pt_bvh_upload looks for faces on the same three points and walks such a scene on the
    binary tree: the image is the oracle's."""
    sc0 = cornell_like.scene
    f, m = sc0.faces.copy(), sc0.materials.copy()
    copies, mats = [f], [m]
    for k, tint in enumerate(((0.9, 0.1, 0.1), (0.1, 0.2, 0.9))):
        fk, mk = f.copy(), m.copy()
        fk["material"] += (k + 1) * len(m)
        mk["baseColorFactor"][:, :3] = tint
        copies.append(fk)
        mats.append(mk)
    n = len(f)
    faces = np.empty(3 * n, f.dtype)
    for i in range(3):  # which copy comes first rotates from face to face
        sel = np.flatnonzero(np.arange(n) % 3 == i)
        for j in range(3):
            faces[3 * sel + j] = copies[(i + j) % 3][sel]
    b = util.Bundle(host.Scene.from_arrays(sc0.vertices.copy(), faces, np.concatenate(mats), lights=sc0.lights.copy(),
                                           camera=sc0.camera))
    assert b.bvh.stats["number_leaves"] > n  # the triples are split over leaves
    monkeypatch.delenv("PT_TREE", raising=False)
    r = core.Renderer(0)
    try:
        o = _setup(r, b, 80, 64)
        gpu = r.render(3, 8, 4)
        assert r.walk_tree() == 2  # the upload saw the coincident faces: no tree timing, binary tree
        cpu = o.render(3, 8, 4)
        _assert_images_match(gpu, cpu)
        # without coincident faces the first launch times both trees
        _setup(r, cornell_like, 80, 64)
        assert r.walk_tree() == 0
    finally:
        r.close()
    # Forced onto the four-wide tree, the walk notices candidates at (nearly) the same distance and repeats
    # itself on the binary tree: slower on such a scene, the same image.
    monkeypatch.setenv("PT_TREE", "wide")
    r = core.Renderer(0)
    try:
        _setup(r, b, 80, 64)
        gpu = r.render(3, 8, 4)
        assert r.walk_tree() == 4
        _assert_images_match(gpu, cpu)
        assert r.count_batch(3, 4)["redone_walks"] > 1000
    finally:
        r.close()


def test_one_coincident_face_among_many_is_found(scene_dir):
    """pt_bvh_upload's search for faces on the same three points, on a scene large enough for its threaded
    path: one copy of one face (corners rotated) among 120 000."""
    b0 = util.load_bundle(scenes.write_many_lights_standin(scene_dir, rings=200, segments=300, name="ml_120k"))
    f = b0.scene.faces.copy()
    assert len(f) > 100000
    r = core.Renderer(0)
    try:
        r.upload(b0.scene, b0.bvh, b0.lst)
        assert r.walk_tree() == 0  # nothing found: the first launch will time both trees
        extra = f[54321:54322].copy()
        extra["vertices"][0] = np.roll(extra["vertices"][0], 1)
        b = util.Bundle(host.Scene.from_arrays(b0.scene.vertices.copy(), np.concatenate([f, extra]), b0.scene.materials.copy(),
                                               lights=b0.scene.lights.copy(), camera=b0.scene.camera))
        r.upload(b.scene, b.bvh, b.lst)
        assert r.walk_tree() == 2
    finally:
        r.close()


def test_height_field_of_tiny_triangles(scene_dir, monkeypatch):
    """A height field of triangles a thousandth of the viewing distance long (the shape of BASELINE config 5):
    Moeller-Trumbore accepts a ray that passes close to an edge in BOTH neighbours, so which of two hits at
    (nearly) the same distance wins depends on the order of the tests and on what the boxes culled. The
    kernel's speculative walks notice such pairs and repeat the ray in the reference's order; without that,
    604 pixels of the full-size configuration differed. Both trees, two launches."""
    b = util.load_bundle(scenes.write_synthetic_grid(scene_dir, name="grid_small", side=500, emitters=4))
    assert len(b.scene.faces) == 2 * 500 * 500 + 8
    cpu = None
    for tree in ("binary", "wide"):
        monkeypatch.setenv("PT_TREE", tree)
        r = core.Renderer(0)
        try:
            o = _setup(r, b, 448, 320)
            gpu = r.render(42, 12, 6)
            if cpu is None:
                cpu = o.render(42, 12, 6)
            _assert_images_match(gpu, cpu)
            c = r.count_batch(42, 6)
            assert c["redone_walks"] > 50, c["redone_walks"]  # the case is there (with the start-over switched
            # off, 2 floats of this image differ on the binary tree and 8 on the wide one)
        finally:
            r.close()


def test_grazing_incidence(monkeypatch):
    """ADVICE r1 (kernel.h margins): the distance Moeller-Trumbore computes loses relative accuracy where a ray
    grazes a triangle (cancellation in det), so a grazing hit can land below the best distance while its box lies
    beyond it; the speculative walks cull against other limits than the reference's loop, and nothing derives
    that kCullSlack / kNearHi cover that. The case, built on purpose: a floor 80 units wide seen from 3e-3 above
    it (cosine to the normal 1e-4 .. 4e-4) and lit by a point light 4e-3 above it, a sheet that crosses the
    floor at 1e-5 rad (two candidates at almost the same distance along every ray near the crossing), fences
    1e-5 .. 5e-4 high standing on it, an emitter 2e-3 above it. Both trees, three seeds, two launches each, and
    the primary hits through the product walker."""
    for scene_seed, fences in ((5, 14), (8, 40), (13, 90)):
        b = util.grazing_scene(scene_seed, fences)
        cpu = {}
        for tree in ("wide", "binary"):
            monkeypatch.setenv("PT_TREE", tree)
            r = core.Renderer(0)
            try:
                o = _setup(r, b, 160, 120)
                for seed in (3, 11, 2024):
                    gpu = r.render(seed, 12, 6)
                    if seed not in cpu:
                        cpu[seed] = o.render(seed, 12, 6)
                    _assert_images_match(gpu, cpu[seed], allow_nan=True)
                ci, ties = _assert_primary_hits_match(r, o, 3, True)
                assert len(np.unique(ci)) >= 6  # floor, sheet, wall, fences and the emitter are in view
                print(f"grazing scene {scene_seed}, {tree} tree: images and primary hits identical; epsilon ties walked again {ties}")
            finally:
                r.close()


def test_flat_shaded_primitive_and_quantised_uvs(renderer, scene_dir):
    """The loader's widening through the kernel: a primitive without NORMAL (flat shaded:
    un-shared vertices carrying the face normal) with normalised-u16 texture coordinates on a textured material."""
    P, N, UV, idx = scenes.blob(7, 9, bump=0.3, seed=4)
    w = scenes.GltfWriter()
    tex = scenes.write_png_rgba8(scene_dir + "/flat_checker.png",
                                 (np.indices((16, 16)).sum(0) % 2)[..., None].astype(np.uint8) * np.uint8([200, 120, 60, 255]) + np.uint8([40, 40, 40, 0]))
    mat = w.material("CHECK", (1, 1, 1, 1), roughness=0.6, base_color_texture=w.texture("flat_checker.png"))
    emit = w.material("EMIT", (0, 0, 0, 1), emissive=(1, 1, 1), emissive_strength=6)
    w.doc.setdefault("extensionsUsed", []).append("KHR_materials_emissive_strength")
    p = w.primitive(P, N, idx, mat, UV)
    del p["attributes"]["NORMAL"]
    q = np.round(np.clip(UV, 0, 1) * 65535).astype(np.uint16)
    view = w._view(q.tobytes(), 34962)
    w.doc["accessors"].append({"bufferView": view, "componentType": 5123, "count": len(q), "type": "VEC2", "normalized": True})
    p["attributes"]["TEXCOORD_0"] = len(w.doc["accessors"]) - 1
    w.node("Blob", mesh=w.mesh("Blob", [p]))
    w.node("Light", mesh=w.mesh("Light", [scenes.unit_plane(w, emit)]), translation=(0, 0, 3), rotation=(1, 0, 0, 0))
    w.node("Camera", camera=w.camera(0.7), translation=(0, 0, 5))
    b = util.load_bundle(w.write(scene_dir, "flat_quantised"))
    assert len(b.scene.vertices) == len(idx) + 4
    o = _setup(renderer, b, 72, 64, clear=(0.05, 0.05, 0.08))
    gpu = renderer.render(6, 10, 5)
    _assert_images_match(gpu, o.render(6, 10, 5))
    assert gpu.mean() > 0.02


def test_two_pixels_per_lane_tiles(cornell_like, monkeypatch):
    """Large images use 8x16 tiles whose 128 pixels are swapped between the 64 lanes every few
    samples (PT_TILE_ROWS=16 forces that kernel at a small size): ragged right and top edges, more
    samples than one turn, several launches."""
    monkeypatch.setenv("PT_TILE_ROWS", "16")
    r = core.Renderer(0)
    try:
        w, h = 100, 70
        o = _setup(r, cornell_like, w, h)
        gpu = r.render(21, 26, 12)  # launches of 12, 12, 2 samples
        cpu = o.render(21, 26, 12)
        _assert_images_match(gpu, cpu)
    finally:
        r.close()


def test_reconfigure_between_sizes_that_share_a_tile_count(cornell_like, monkeypatch):
    """16x24 and 24x16 have the same number of 8x8 tiles (6) but map 8x16 tiles differently (2x2 against
    3x1): the launch order recorded for one is no permutation for the other, so pt_configure must drop it
    whenever the size changes. Also with one pixel per lane, 48x16 then 16x48."""
    for rows, sizes in (("16", ((16, 24), (24, 16))), ("8", ((48, 16), (16, 48)))):
        monkeypatch.setenv("PT_TILE_ROWS", rows)
        r = core.Renderer(0)
        try:
            r.upload(cornell_like.scene, cornell_like.bvh, cornell_like.lst)
            for w, h in sizes:
                r.configure(w, h)
                o = ob.Oracle(cornell_like.scene, cornell_like.bvh, cornell_like.lst, w, h)
                gpu = r.render(5, 16, 8)  # launches of 8 samples: the tile costs are recorded and ranked
                _assert_images_match(gpu, o.render(5, 16, 8))
        finally:
            r.close()


def test_combine_is_ordered_after_the_render_launches(scene_dir):
    """the scaling + reduce of render.combine must see the finished accumulator. Several
    1024x1024 batches (tens of ms of kernels in flight) on the side stream render_sharded uses, then
    combine(acc, local, 2 * local): the result is exactly half the batch-loop image. With the launches on a
    stream torch does not order against, the scaling would run while the kernels are still blending."""
    import torch
    from pathtracer_amd import render
    b = util.load_bundle(scenes.write_many_lights_standin(scene_dir, rings=60, segments=96, name="ml_order"))
    w = h = 1024
    r = core.Renderer(0)
    try:
        r.upload(b.scene, b.bvh, b.lst)
        r.configure(w, h)
        want = r.render(42, 24, 8)
        device = torch.device("cuda", 0)
        side = torch.cuda.Stream(device)
        with torch.cuda.stream(side):
            acc = torch.zeros(w * h * 3, dtype=torch.float32, device=device)
            local = render.render_local(r, acc, 42, render.plan_batches(24, 8), side.cuda_stream)
            render.combine(acc, local, 2 * local)
        side.synchronize()
        got = acc.cpu().numpy().reshape(h, w, 3)
        assert local == 24
        assert np.array_equal(util.bits(got), util.bits(want * np.float32(0.5)))
        # the default stream's handle (0) is refused instead of being taken for "the context's own stream"
        with pytest.raises(ValueError):
            r.render_batch_into(acc.data_ptr(), 1, 0, 1, stream=0)
        # and the whole of render_sharded, called from the default stream
        img = render.render_sharded(r, w, h, 42, 24, 8)
        torch.cuda.synchronize()
        assert np.array_equal(util.bits(img.cpu().numpy()), util.bits(want))
    finally:
        r.close()


def test_tiny_and_ragged_images(renderer, cornell_like):
    """Images smaller than one tile and one pixel wide or high; a launch that continues a running mean from a
    non-zero previous sample count into a pre-filled accumulator (renderer.cu:398-404); seed 2^31 - 1 (the host
    loop's seed + 1 then wraps like the reference's int)."""
    for w, h in ((1, 1), (3, 2), (9, 1), (1, 17), (8, 8), (65, 9)):
        o = _setup(renderer, cornell_like, w, h)
        gpu = renderer.render(-3, 5, 2)
        _assert_images_match(gpu, o.render(-3, 5, 2))
    o = _setup(renderer, cornell_like, 24, 16)
    renderer.clear()
    renderer.render_batch(7, 0, 3)
    renderer.render_batch(2 ** 31 - 1, 3, 4)
    cpu, _ = o.render_batch(7, 0, 3)
    o.render_batch(2 ** 31 - 1, 3, 4, cpu)
    _assert_images_match(renderer.read_image(), cpu)


def test_counters_match_oracle(renderer, cornell_like):
    """Reference-equivalent ray count, samples, shaded hits and bounces are integers that must
    agree exactly; the kernel walks fewer trees than the reference traces rays."""
    o = _setup(renderer, cornell_like, 64, 64)
    g = renderer.count_batch(42, 4)
    _, c = o.render_batch(42, 0, 4)
    for k in ("samples", "rays", "shaded_hits", "bounces"):
        assert g[k] == c[k], (k, g[k], c[k])
    assert g["traversals"] < c["rays"]
    assert g["inner_visits"] < c["inner_visits"]


def test_no_lights_scene(renderer, scene_dir):
    """Empty light table: no NEE walk, paths end by Russian roulette or by escaping."""
    b = util.load_bundle(scenes.write_cornell_like(scene_dir, name="nolight"))
    b.lst.nodes = b.lst.nodes[:0]
    o = _setup(renderer, b, 32, 32, clear=(0.5, 0.5, 0.5))
    renderer.clear()
    renderer.render_batch(1, 0, 4)
    cpu, _ = o.render_batch(1, 0, 4)
    _assert_images_match(renderer.read_image(), cpu)


def test_api_errors(renderer, cornell_like):
    r = core.Renderer(0)
    with pytest.raises(core.CoreError):
        r.render_batch(1, 0, 1)  # nothing uploaded
    r.upload(cornell_like.scene, cornell_like.bvh, cornell_like.lst)
    with pytest.raises(core.CoreError):
        r.configure(0, 10)
    r.configure(8, 8)
    with pytest.raises(core.CoreError):
        r.render_batch(1, 0, 0)
    # an upload that fails leaves no usable tree behind: the next launch is refused, not run on stale data
    r.render_batch(1, 0, 1)
    bad = np.ascontiguousarray(cornell_like.bvh.nodes).copy()
    bad["leftChild"][0] = len(bad) + 7
    lib = r._lib
    assert lib.pt_bvh_upload(r._ctx, bad.ctypes.data, len(bad), np.ascontiguousarray(cornell_like.bvh.indices).ctypes.data,
                             len(cornell_like.bvh.indices)) != 0
    with pytest.raises(core.CoreError, match="pt_bvh_upload"):
        r.render_batch(1, 0, 1)
    r.upload(cornell_like.scene, cornell_like.bvh, cornell_like.lst)
    r.render_batch(1, 0, 1)
    r.close()
    with pytest.raises(core.CoreError):
        core.Renderer(99)
