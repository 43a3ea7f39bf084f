"""Host pipeline objects (thin wrappers over libpthost.so).

Mirrors the reference's host-side classes for this path: `Scene` (src/scene.h:149-167),
`BVH` (src/bvh.h:27-39), `LST` (src/lst.h:18-27) and `write_image` (src/image.h). All work
happens in C++; these classes only hold the handles and expose the arrays as numpy views.
"""
import ctypes as C

import numpy as np

from . import _ffi


class HostError(RuntimeError):
    pass


def _check(rc):
    if rc:
        raise HostError(_ffi.host_lib().pth_last_error().decode("utf-8", "replace"))


def _view(ptr, count, dtype):
    if not ptr or count == 0:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (count * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=count)


def make_config(**kw):
    """A pth_config with the given fields set (strings are encoded)."""
    cfg = _ffi.HostConfig()
    for k, v in kw.items():
        if isinstance(v, str):
            v = v.encode()
        elif isinstance(v, (tuple, list)) and len(v) == 3:
            v = _ffi.Vec3(*[float(c) for c in v])
        setattr(cfg, k, v)
    return cfg


class Scene:
    """Owning handle on a pth_scene. `desc` is the pt_scene_desc handed to the device side."""

    def __init__(self, handle, keepalive=None):
        self._h = handle
        self._keep = keepalive
        self._lib = _ffi.host_lib()
        self.desc = self._lib.pth_scene_get_desc(handle).contents

    @classmethod
    def load_gltf(cls, path, clear_texture=None, default_camera=None):
        """Scene::read_gltf (src/scene.cu:839). clear_texture = --world-clear-color-texture."""
        lib = _ffi.host_lib()
        kw = {}
        if clear_texture:
            kw["world_clear_color_texture"] = clear_texture
        if default_camera:
            kw.update(default_camera_position=default_camera["position"], default_camera_target=default_camera["target"],
                      default_camera_updir=default_camera["updir"], default_camera_yfov=default_camera["yfov"])
        cfg = make_config(**kw)
        h = C.c_void_p()
        _check(lib.pth_scene_load_gltf(str(path).encode(), C.byref(cfg), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, vertices, faces, materials, lights=None, camera=None, textures=(), clear_texture=-1):
        """Build from numpy arrays of the record dtypes in _ffi (deep-copied by the library).

        textures: list of dicts {pixels: ndarray (h,w,4) float32|uint8, wrapS, wrapT, filter}.
        """
        lib = _ffi.host_lib()
        vertices = np.ascontiguousarray(vertices, dtype=_ffi.VERTEX_DTYPE)
        faces = np.ascontiguousarray(faces, dtype=_ffi.FACE_DTYPE)
        materials = np.ascontiguousarray(materials, dtype=_ffi.MATERIAL_DTYPE)
        lights = np.ascontiguousarray(lights if lights is not None else np.zeros(0, _ffi.LIGHT_DTYPE), dtype=_ffi.LIGHT_DTYPE)
        d = _ffi.SceneDesc()
        d.vertices, d.vertex_count = vertices.ctypes.data, len(vertices)
        d.faces, d.face_count = faces.ctypes.data, len(faces)
        d.materials, d.material_count = materials.ctypes.data, len(materials)
        d.lights, d.light_count = lights.ctypes.data, len(lights)
        keep = []
        tex_arr = (_ffi.Texture * max(1, len(textures)))()
        for i, t in enumerate(textures):
            px = np.ascontiguousarray(t["pixels"])
            assert px.ndim == 3 and px.shape[2] == 4 and px.dtype in (np.float32, np.uint8)
            keep.append(px)
            tex_arr[i].width, tex_arr[i].height, tex_arr[i].channels = px.shape[1], px.shape[0], 4
            tex_arr[i].isFloat = 1 if px.dtype == np.float32 else 0
            tex_arr[i].wrapS = t.get("wrapS", 0)
            tex_arr[i].wrapT = t.get("wrapT", 0)
            tex_arr[i].filter = t.get("filter", 1)
            tex_arr[i].pixels = px.ctypes.data
        d.textures = C.cast(tex_arr, C.POINTER(_ffi.Texture))
        d.texture_count = len(textures)
        d.clearTexture = clear_texture
        if camera is not None:
            d.camera = _ffi.Camera(_ffi.Vec3(*camera["position"]), _ffi.Vec3(*camera["target"]),
                                   _ffi.Vec3(*camera["updir"]), float(camera["yfov"]))
        h = C.c_void_p()
        _check(lib.pth_scene_from_desc(C.byref(d), C.byref(h)))
        return cls(h)

    @property
    def vertices(self):
        return _view(self.desc.vertices, self.desc.vertex_count, _ffi.VERTEX_DTYPE)

    @property
    def faces(self):
        return _view(self.desc.faces, self.desc.face_count, _ffi.FACE_DTYPE)

    @property
    def materials(self):
        return _view(self.desc.materials, self.desc.material_count, _ffi.MATERIAL_DTYPE)

    @property
    def lights(self):
        return _view(self.desc.lights, self.desc.light_count, _ffi.LIGHT_DTYPE)

    @property
    def camera(self):
        c = self.desc.camera
        return {"position": (c.position.x, c.position.y, c.position.z), "target": (c.target.x, c.target.y, c.target.z),
                "updir": (c.updir.x, c.updir.y, c.updir.z), "yfov": c.yfov}

    def texture(self, i):
        t = self.desc.textures[i]
        dtype = np.dtype("<f4") if t.isFloat else np.dtype("u1")
        return _view(t.pixels, t.width * t.height * 4, dtype).reshape(t.height, t.width, 4), t

    def close(self):
        if self._h:
            self._lib.pth_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BVH:
    """BVH::build (src/bvh.cu:288). `nodes` / `indices` are views into the C arrays."""

    def __init__(self, scene):
        self._lib = _ffi.host_lib()
        self._b = _ffi.HostBvh()
        self._scene = scene
        _check(self._lib.pth_bvh_build(C.byref(scene.desc), C.byref(self._b)))

    @property
    def node_count(self):
        return self._b.node_count

    @property
    def node_capacity(self):
        return self._b.node_capacity

    @property
    def nodes(self):
        return _view(self._b.nodes, self._b.node_count, _ffi.BVH_NODE_DTYPE)

    @property
    def indices(self):
        return _view(self._b.indices, self._b.index_count, np.dtype("<u4"))

    @property
    def stats(self):
        b = self._b
        return {"node_count": b.node_count, "node_capacity": b.node_capacity, "number_leaves": b.number_leaves,
                "total_skipped_faces": b.total_skipped_faces, "average_leaf_size": b.average_leaf_size,
                "max_depth": b.max_depth}

    def close(self):
        if self._b.nodes:
            self._lib.pth_bvh_free(C.byref(self._b))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class LST:
    """LST::build (src/lst.cu:6): flat list of emissive faces then punctual lights."""

    def __init__(self, scene):
        lib = _ffi.host_lib()
        p = C.c_void_p()
        n = C.c_uint32()
        _check(lib.pth_lst_build(C.byref(scene.desc), C.byref(p), C.byref(n)))
        self.nodes = _view(p.value, n.value, _ffi.LST_NODE_DTYPE).copy()
        lib.pth_free(p)


def calculate_tangents(vertices, faces, first_face=0, num_faces=None):
    """calculate_tangents (src/scene.cu:371) in place on a VERTEX_DTYPE array."""
    lib = _ffi.host_lib()
    assert vertices.dtype == _ffi.VERTEX_DTYPE and faces.dtype == _ffi.FACE_DTYPE
    assert vertices.flags.c_contiguous and faces.flags.c_contiguous
    n = len(faces) - first_face if num_faces is None else num_faces
    _check(lib.pth_calculate_tangents(vertices.ctypes.data, len(vertices), faces.ctypes.data, first_face, n))


def load_hdr(path):
    lib = _ffi.host_lib()
    w, h = C.c_int(), C.c_int()
    p = C.POINTER(C.c_float)()
    _check(lib.pth_load_hdr(str(path).encode(), C.byref(w), C.byref(h), C.byref(p)))
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    lib.pth_free(p)
    return out


def load_png(path):
    lib = _ffi.host_lib()
    w, h = C.c_int(), C.c_int()
    p = C.POINTER(C.c_uint8)()
    _check(lib.pth_load_png(str(path).encode(), C.byref(w), C.byref(h), C.byref(p)))
    out = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    lib.pth_free(p)
    return out


def tonemap_rgb8(linear):
    """The 8-bit RGB buffer write_image hands to the PNG encoder (src/image.cu:72-99)."""
    lib = _ffi.host_lib()
    linear = np.ascontiguousarray(linear, dtype=np.float32)
    h, w = linear.shape[:2]
    out = np.zeros((h, w, 3), np.uint8)
    _check(lib.pth_tonemap_rgb8(linear.ctypes.data, w, h, out.ctypes.data))
    return out


def write_image(linear, path):
    lib = _ffi.host_lib()
    linear = np.ascontiguousarray(linear, dtype=np.float32)
    h, w = linear.shape[:2]
    _check(lib.pth_write_image(linear.ctypes.data, w, h, str(path).encode()))
