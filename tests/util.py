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


def golden_cornell():
    """The reference's cornell.gltf after the host pipeline, stored as derived arrays
    (tests/golden/cornell_scene.npz, made by tests/golden/make_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "cornell_scene.npz"))
    cam = {"position": tuple(z["camera"][0:3]), "target": tuple(z["camera"][3:6]), "updir": tuple(z["camera"][6:9]),
           "yfov": float(z["camera"][9])}
    sc = host.Scene.from_arrays(z["vertices"].view(_ffi.VERTEX_DTYPE).reshape(-1),
                                z["faces"].view(_ffi.FACE_DTYPE).reshape(-1),
                                z["materials"].view(_ffi.MATERIAL_DTYPE).reshape(-1), camera=cam)
    return Bundle(sc)


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
