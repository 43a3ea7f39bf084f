"""Shared helpers for the test suite."""
import functools
import os

import numpy as np

from pathtracer_amd import _ffi, host, scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Bundle:
    """Scene + BVH + LST built by the host pipeline."""

    def __init__(self, scene):
        self.scene = scene
        self.bvh = host.BVH(scene)
        self.lst = host.LST(scene)


def load_bundle(path, **kw):
    return Bundle(host.Scene.load_gltf(path, **kw))


def golden_scene(name, env=False):
    """One of the reference's own scenes (cornell, dragon, glass_mirror) after the host pipeline, stored as
    derived arrays (tests/golden/<name>_scene.npz, made by tests/golden/make_golden.py where the reference is
    mounted), so that the BASELINE configurations on real assets run on the GPU box, which has no reference.
    env=True adds BASELINE config 4's environment map as the clear texture (rose_garden_env.npz: the
    reference's sunny_rose_garden_1k.hdr decoded and point-sampled 4x DOWN to 256x128: the real texels, not the
    real resolution), with the sampler state the reference gives an environment map: linear filter, mirrored
    repeat on both axes (scene.cu:928-936). env="sky1k": a procedural map at the real 1024x512 resolution."""
    z = np.load(os.path.join(GOLDEN, f"{name}_scene.npz"))
    cam = {"position": tuple(z["camera"][0:3]), "target": tuple(z["camera"][3:6]), "updir": tuple(z["camera"][6:9]),
           "yfov": float(z["camera"][9])}
    textures, clear = (), -1
    if env == "sky1k":
        # an environment map at config 4's FULL resolution (1024x512, the size of sunny_rose_garden_1k.hdr): a procedural
        # Radiance file written here (no asset copied) and read back through the product's own .hdr decoder
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            px = host.load_hdr(scenes.write_sky_hdr(os.path.join(d, "sky1k.hdr"), 1024, 512))
        assert px.shape == (512, 1024, 4) and px.dtype == np.float32
        textures, clear = ({"pixels": px, "wrapS": 2, "wrapT": 2, "filter": 1},), 0
    elif env:
        rgb = np.load(os.path.join(GOLDEN, "rose_garden_env.npz"))["rgb"]
        px = np.ones(rgb.shape[:2] + (4,), np.float32)
        px[..., :3] = rgb
        textures, clear = ({"pixels": px, "wrapS": 2, "wrapT": 2, "filter": 1},), 0
    lights = z["lights"].view(_ffi.LIGHT_DTYPE).reshape(-1) if "lights" in z.files else None
    sc = host.Scene.from_arrays(z["vertices"].view(_ffi.VERTEX_DTYPE).reshape(-1),
                                z["faces"].view(_ffi.FACE_DTYPE).reshape(-1),
                                z["materials"].view(_ffi.MATERIAL_DTYPE).reshape(-1), lights=lights, camera=cam,
                                textures=textures, clear_texture=clear)
    return Bundle(sc)


def golden_cornell():
    return golden_scene("cornell")


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def count_different(a, b):
    """Floats that differ in bits, where a NaN on both sides counts as equal (the sign and payload of a NaN
    that 0/0 produces differ between x86 and gfx950; which pixels are NaN must not)."""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return int(((bits(a) != bits(b)) & ~(np.isnan(a) & np.isnan(b))).sum())


def grazing_scene(seed=5, n_slabs=14):
    """A large floor seen and lit at a cosine of about 1e-4, thin occluders standing just behind where the rays
    meet it, and two huge triangles at a tiny angle to it (near-coincident competitors)."""
    rng = np.random.RandomState(seed)
    P, N, F, M = [], [], [], []
    def tri(a, b, c, mat):
        n = np.cross(np.subtract(b, a), np.subtract(c, a)).astype(np.float64)
        n = n / np.linalg.norm(n)
        i = len(P)
        P.extend([a, b, c]); N.extend([n, n, n]); F.append((i, i + 1, i + 2)); M.append(mat)
    def quad(a, b, c, d, mat):
        tri(a, b, c, mat); tri(a, c, d, mat)
    quad((-40, -40, 0), (40, -40, 0), (40, 40, 0), (-40, 40, 0), 0)            # the floor
    quad((-40, -40, -2e-4), (40, -40, 6e-4), (40, 40, 6e-4), (-40, 40, -2e-4), 1)  # a sheet crossing it at 1e-5 rad
    quad((35, -40, -1), (35, 40, -1), (35, 40, 1), (35, -40, 1), 2)            # back wall
    for k in range(n_slabs):                                                     # thin fences on the floor
        x = rng.uniform(-22, 30); y0 = rng.uniform(-3, -0.004); w = rng.uniform(0.003, 3.5); hgt = 10 ** rng.uniform(-5, -3.3)
        quad((x, y0, -hgt), (x, y0 + w, -hgt), (x, y0 + w, hgt), (x, y0, hgt), 1 + k % 2)
    quad((10, -3, 0.5), (14, -3, 0.5), (14, 3, 0.5), (10, 3, 0.5), 3)          # emitter above
    quad((-20, -1, 2e-3), (-19, -1, 2e-3), (-19, 1, 2e-3), (-20, 1, 2e-3), 3)   # emitter lying almost in the floor plane
    P = np.array(P, np.float32); N = np.array(N, np.float32)
    v = np.zeros(len(P), _ffi.VERTEX_DTYPE)
    v["position"] = P; v["normal"] = N
    t = np.cross(N, (0.3, 0.5, 0.8)); t /= np.linalg.norm(t, axis=1, keepdims=True)
    v["tangent"][:, :3] = t; v["tangent"][:, 3] = 1
    v["texcoord0"][:, :2] = rng.rand(len(P), 2)
    f = np.zeros(len(F), _ffi.FACE_DTYPE)
    f["vertices"] = np.array(F, np.uint32); f["vertexCount"] = 3; f["material"] = np.array(M, np.uint32); f["shading"] = 1
    m = np.zeros(4, _ffi.MATERIAL_DTYPE)
    for i, (name, col, rough, emis) in enumerate((("FLOOR", (0.8, 0.8, 0.8, 1), 0.5, 0), ("SHEET", (0.9, 0.3, 0.2, 1), 0.3, 0),
                                                   ("WALL", (0.2, 0.4, 0.9, 1), 0.7, 0), ("EMIT", (0, 0, 0, 1), 0.5, 6.0))):
        m[i]["name"] = name.encode(); m[i]["baseColorFactor"] = col; m[i]["roughnessFactor"] = rough; m[i]["metallicFactor"] = 0.0
        m[i]["emissive"] = (emis,) * 3; m[i]["baseColorTexture"] = m[i]["normalTexture"] = m[i]["metallicRoughnessTexture"] = -1
        m[i]["alphaCutoff"] = 0.5; m[i]["ior"] = 1.3
    lights = np.zeros(1, _ffi.LIGHT_DTYPE)
    lights[0]["type"] = 1; lights[0]["color"] = (1, 1, 1); lights[0]["position"] = (-35, 2, 4e-3); lights[0]["intensity"] = 4000.0
    lights[0]["direction"] = (0, 0, -1)
    cam = {"position": (-30.0, 0.0, 3e-3), "target": (0.0, 0.0, 0.0), "updir": (0.0, 0.0, 1.0), "yfov": 6e-4}
    return Bundle(host.Scene.from_arrays(v, f, m, lights=lights, camera=cam))
