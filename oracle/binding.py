"""ctypes binding of liboracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by
the pathtracer_amd package. See oracle/oracle.h for what the oracle is and what pins it.
"""
import ctypes as C
import os

import numpy as np

from pathtracer_amd import _ffi

_HERE = os.path.dirname(os.path.abspath(__file__))


class Inputs(C.Structure):  # orc_inputs
    _fields_ = [
        ("scene", C.POINTER(_ffi.SceneDesc)),
        ("nodes", C.c_void_p), ("node_count", C.c_uint32),
        ("indices", C.c_void_p), ("index_count", C.c_uint32),
        ("lst", C.c_void_p), ("lst_count", C.c_uint32),
        ("cfg", _ffi.RenderConfig),
    ]


class NonfiniteDiag(C.Structure):  # orc_nonfinite_diag
    _fields_ = [(n, C.c_int32) for n in ("found", "where", "sample", "bounce", "face", "material")] + \
               [(n, C.c_float) for n in ("roughness", "metallic", "h_z", "pdf")] + \
               [(n, C.c_float * 3) for n in ("wo_ts", "w_ts", "f_cos", "tangent", "bitangent", "normal", "ray_dir")]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if hasattr(getattr(self, n), "__len__") else getattr(self, n)) for n, _ in self._fields_}


class MarginStats(C.Structure):  # orc_margin_stats
    _fields_ = [("rays", C.c_uint64), ("leaf_looks", C.c_uint64), ("hits", C.c_uint64),
                ("entry_behind", C.c_uint64), ("exit_in_front", C.c_uint64),
                ("max_entry_behind", C.c_double), ("max_exit_in_front", C.c_double),
                ("culled_by_distance", C.c_uint64), ("max_culled_gap", C.c_double),
                ("missed_box", C.c_uint64), ("max_rim_of_missed", C.c_double),
                ("missed_box_inf_rinv", C.c_uint64), ("max_rim_of_missed_finite", C.c_double), ("max_t_of_missed_finite", C.c_double),
                ("rays_with_two_candidates", C.c_uint64), ("gap_below", C.c_uint64 * 16)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise OSError(f"{path} missing: run `make -C oracle`")
        L = C.CDLL(path)
        L.orc_render_batch.argtypes = [C.POINTER(Inputs), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                       C.POINTER(_ffi.Counters)]
        L.orc_render_rows.argtypes = [C.POINTER(Inputs), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(_ffi.Counters)]
        L.orc_rng_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
        L.orc_rng_stream.restype = None
        L.orc_rng_set_scramble.argtypes = [C.c_void_p]
        L.orc_rng_set_scramble.restype = None
        L.orc_rng_state.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_rng_state.restype = None
        L.orc_rng_sequence_jump_matrix.argtypes = [C.c_int, C.c_void_p]
        L.orc_rng_sequence_jump_matrix.restype = None
        L.orc_primary_hits.argtypes = [C.POINTER(Inputs), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_find_nonfinite.argtypes = [C.POINTER(Inputs), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(NonfiniteDiag)]
        L.orc_margin_probe.argtypes = [C.POINTER(Inputs), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(MarginStats)]
        L.orc_tonemap_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_sample_texture.argtypes = [C.POINTER(_ffi.Texture), C.c_float, C.c_float, C.c_void_p]
        L.orc_sample_texture.restype = None
        L.orc_logf.argtypes = [C.c_float]
        L.orc_logf.restype = C.c_float
        L.orc_sincosf.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_sincosf.restype = None
        L.orc_atan2f.argtypes = [C.c_float, C.c_float]
        L.orc_atan2f.restype = C.c_float
        L.orc_acosf.argtypes = [C.c_float]
        L.orc_acosf.restype = C.c_float
        _lib = L
    return _lib


class Oracle:
    """The oracle bound to one (scene, bvh, lst, config)."""

    def __init__(self, scene, bvh, lst, width, height, exposure=0.0, clear_color=(0.0, 0.0, 0.0)):
        self.scene, self.bvh, self.lst = scene, bvh, lst  # keep alive
        self._nodes = np.ascontiguousarray(bvh.nodes)
        self._indices = np.ascontiguousarray(bvh.indices)
        self._lst = np.ascontiguousarray(lst.nodes)
        self.width, self.height = width, height
        i = Inputs()
        i.scene = C.pointer(scene.desc)
        i.nodes, i.node_count = self._nodes.ctypes.data, len(self._nodes)
        i.indices, i.index_count = self._indices.ctypes.data, len(self._indices)
        i.lst, i.lst_count = self._lst.ctypes.data, len(self._lst)
        i.cfg = _ffi.RenderConfig(width, height, float(exposure), _ffi.Vec3(*[float(c) for c in clear_color]))
        self.inputs = i

    def render_batch(self, seed, previous_samples, current_samples, img=None, threads=0, rows=None):
        """render_kernel over the image (or rows [y0,y1)); returns (img, counters)."""
        if img is None:
            img = np.zeros((self.height, self.width, 3), np.float32)
        assert img.dtype == np.float32 and img.flags.c_contiguous
        ctr = _ffi.Counters()
        y0, y1 = rows if rows is not None else (0, self.height)
        rc = lib().orc_render_rows(C.byref(self.inputs), seed, previous_samples, current_samples, img.ctypes.data,
                                   y0, y1, threads, C.byref(ctr))
        if rc:
            raise RuntimeError("oracle render failed")
        return img, ctr.as_dict()

    def render(self, seed, samples, samples_every_update, threads=0):
        """The batch loop of main.cu:101-133 (seed + 1 per batch)."""
        img = np.zeros((self.height, self.width, 3), np.float32)
        done = 0
        while done < samples:
            cur = min(samples_every_update, samples - done)
            self.render_batch(seed, done, cur, img, threads)
            done += cur
            seed += 1
        return img

    def find_nonfinite(self, seed, current_samples, px, py):
        """Where pixel (px, py) of that batch first turns non-finite (None if it does not)."""
        d = NonfiniteDiag()
        return d.as_dict() if lib().orc_find_nonfinite(C.byref(self.inputs), seed, current_samples, px, py, C.byref(d)) else None

    def margin_probe(self, seed, current_samples, threads=0, rows=None):
        """orc_margin_probe: how far box and triangle arithmetic disagree along the rays of this render."""
        m = MarginStats()
        y0, y1 = rows if rows is not None else (0, self.height)
        if lib().orc_margin_probe(C.byref(self.inputs), seed, current_samples, y0, y1, threads, C.byref(m)):
            raise RuntimeError("oracle margin probe failed")
        d = {n: getattr(m, n) for n, _ in MarginStats._fields_ if n != "gap_below"}
        d["gap_below"] = list(m.gap_below)
        return d

    def primary_hits(self, seed, jitter=True):
        ids = np.zeros((self.height, self.width), np.int32)
        dist = np.zeros((self.height, self.width), np.float32)
        rc = lib().orc_primary_hits(C.byref(self.inputs), seed, 1 if jitter else 0, ids.ctypes.data, dist.ctypes.data)
        if rc:
            raise RuntimeError("oracle primary_hits failed")
        return ids, dist


def tonemap_rgb8(linear):
    """The oracle's restatement of write_image's 8-bit buffer (src/image.cu:72-99)."""
    linear = np.ascontiguousarray(linear, dtype=np.float32)
    h, w = linear.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    if lib().orc_tonemap_rgb8(linear.ctypes.data, w, h, out.ctypes.data):
        raise ValueError("non-finite pixel")
    return out


def rng_stream(seed, subsequence, n):
    out = np.zeros(n, np.uint32)
    lib().orc_rng_stream(seed & 0xFFFFFFFFFFFFFFFF, subsequence, out.ctypes.data, n)
    return out


def set_scramble(constants=None):
    """Test hook: the four seed-scramble constants (None = cuRAND's, as recalled)."""
    if constants is None:
        lib().orc_rng_set_scramble(None)
    else:
        a = np.asarray(constants, np.uint32)
        assert a.shape == (4,)
        lib().orc_rng_set_scramble(a.ctypes.data)


def rng_state(seed, subsequence):
    out = np.zeros(6, np.uint32)
    lib().orc_rng_state(seed & 0xFFFFFFFFFFFFFFFF, subsequence, out.ctypes.data)
    return out


def sequence_jump_matrix(k):
    out = np.zeros(800, np.uint32)
    lib().orc_rng_sequence_jump_matrix(k, out.ctypes.data)
    return out
