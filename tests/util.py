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
    reference's sunny_rose_garden_1k.hdr decoded and point-sampled down to 256x128), with the sampler state
    the reference gives an environment map: linear filter, mirrored repeat on both axes (scene.cu:928-936)."""
    z = np.load(os.path.join(GOLDEN, f"{name}_scene.npz"))
    cam = {"position": tuple(z["camera"][0:3]), "target": tuple(z["camera"][3:6]), "updir": tuple(z["camera"][6:9]),
           "yfov": float(z["camera"][9])}
    textures, clear = (), -1
    if env:
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
