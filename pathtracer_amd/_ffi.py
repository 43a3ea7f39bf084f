"""ctypes view of include/pt_types.h, include/pthost.h and include/ptcore.h.

Plumbing only: record layouts (as ctypes structures and numpy dtypes of the same
byte layout) and the loaders for the two in-tree shared libraries. No compute.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
REPO_ROOT = os.path.dirname(_HERE)


class Vec3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("position", Vec3), ("target", Vec3), ("updir", Vec3), ("yfov", C.c_float)]


class Texture(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("channels", C.c_int32), ("isFloat", C.c_int32),
        ("wrapS", C.c_int32), ("wrapT", C.c_int32), ("filter", C.c_int32), ("_pad", C.c_int32),
        ("pixels", C.c_void_p),
    ]


class SceneDesc(C.Structure):
    _fields_ = [
        ("vertices", C.c_void_p), ("vertex_count", C.c_uint32),
        ("faces", C.c_void_p), ("face_count", C.c_uint32),
        ("lights", C.c_void_p), ("light_count", C.c_uint32),
        ("materials", C.c_void_p), ("material_count", C.c_uint32),
        ("textures", C.POINTER(Texture)), ("texture_count", C.c_uint32),
        ("camera", Camera),
        ("clearTexture", C.c_int32),
    ]


class RenderConfig(C.Structure):
    _fields_ = [("resolution_x", C.c_int32), ("resolution_y", C.c_int32),
                ("output_exposure", C.c_float), ("world_clear_color", Vec3)]


class HostConfig(C.Structure):  # pth_config
    _fields_ = [
        ("resolution_x", C.c_int), ("resolution_y", C.c_int), ("output_exposure", C.c_float),
        ("samples", C.c_int), ("seed", C.c_int), ("samples_every_update", C.c_int),
        ("world_clear_color", Vec3), ("world_clear_color_texture", C.c_char * 256),
        ("log_level", C.c_int), ("log_stdout", C.c_int),
        ("path_gltf", C.c_char * 256), ("dir_output", C.c_char * 256), ("path_render", C.c_char * 256),
        ("default_camera_position", Vec3), ("default_camera_target", Vec3), ("default_camera_updir", Vec3),
        ("default_camera_yfov", C.c_float),
    ]


class HostBvh(C.Structure):  # pth_bvh
    _fields_ = [
        ("nodes", C.c_void_p), ("node_capacity", C.c_uint32), ("node_count", C.c_uint32),
        ("indices", C.c_void_p), ("index_count", C.c_uint32),
        ("total_skipped_faces", C.c_uint32), ("number_leaves", C.c_uint32),
        ("average_leaf_size", C.c_float), ("max_depth", C.c_int),
    ]


class Counters(C.Structure):  # orc_counters (oracle/oracle.h)
    _fields_ = [(n, C.c_uint64) for n in
                ("samples", "rays", "inner_visits", "tri_tests", "hit_records", "shaded_hits", "bounces")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class CoreCounters(C.Structure):  # pt_counters (include/ptcore.h)
    _fields_ = [(n, C.c_uint64) for n in
                ("samples", "rays", "inner_visits", "tri_tests", "traversals", "shaded_hits", "bounces")] + \
               [("region_iters", C.c_uint64 * 6), ("region_lanes", C.c_uint64 * 6),
                ("region_ticks", C.c_uint64 * 6), ("wave_ticks", C.c_uint64),
                ("node_step_idle", C.c_uint64 * 3), ("redone_walks", C.c_uint64)]
    REGIONS = ("loop", "inner", "leaf", "main", "nee", "bsdf")

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_[:7]}
        d["lane_utilisation"] = {r: round(self.region_lanes[i] / (64.0 * self.region_iters[i]), 4) if self.region_iters[i] else None
                                 for i, r in enumerate(self.REGIONS)}
        d["wave_iterations"] = {r: int(self.region_iters[i]) for i, r in enumerate(self.REGIONS)}
        d["redone_walks"] = int(self.redone_walks)
        # per-lane executions of the regions (sum of active lanes over the waves' passes)
        d["leaf_tests"], d["handler_calls"], d["shadow_results"] = (int(self.region_lanes[2]), int(self.region_lanes[3]), int(self.region_lanes[4]))
        if self.region_iters[1]:
            tot = 64.0 * self.region_iters[1]
            d["node_step_lanes"] = {"stepping": round(self.region_lanes[1] / tot, 4),
                                    "blocked_on_leaves": round(self.node_step_idle[0] / tot, 4),
                                    "parked_leaf_only": round(self.node_step_idle[1] / tot, 4),
                                    "awaiting_handler": round(self.node_step_idle[2] / tot, 4)}
        if self.wave_ticks:
            names = ("walk_slices", "node_phases", "leaf_phases", "vertex_handler", "shadow_results", "rejection_loops")
            d["time_share"] = {nm: round(self.region_ticks[i] / self.wave_ticks, 4) for i, nm in enumerate(names)}
            d["wave_ticks"] = int(self.wave_ticks)
        return d


class GroupBatch(C.Structure):  # pt_group_batch (include/ptcore.h)
    _fields_ = [("index", C.c_int), ("rank", C.c_int), ("seed", C.c_int), ("previous_samples", C.c_int), ("current_samples", C.c_int)]


VEC3 = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
VEC4 = VEC3 + [("w", "<f4")]
VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("texcoord0", "<f4", 3), ("tangent", "<f4", 4)])
FACE_DTYPE = np.dtype([("vertices", "<u4", 3), ("vertexCount", "<u4"), ("material", "<u4"),
                       ("normal", "<f4", 3), ("tangent", "<f4", 4), ("shading", "<u4")])
MATERIAL_DTYPE = np.dtype([("name", "S32"), ("baseColorFactor", "<f4", 4),
                           ("baseColorTexture", "<i4"), ("normalTexture", "<i4"), ("metallicRoughnessTexture", "<i4"),
                           ("emissive", "<f4", 3), ("specular", "<f4"), ("metallicFactor", "<f4"),
                           ("roughnessFactor", "<f4"), ("ior", "<f4"), ("transmission", "<f4"),
                           ("alphaMode", "<u4"), ("alphaCutoff", "<f4")])
LIGHT_DTYPE = np.dtype([("type", "<u4"), ("color", "<f4", 3), ("position", "<f4", 3), ("direction", "<f4", 3),
                        ("intensity", "<f4")])
BVH_NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("leftChild", "<u4"), ("rightChild", "<u4"),
                           ("start", "<u4"), ("end", "<u4")])
LST_NODE_DTYPE = np.dtype([("type", "<u4"), ("index", "<u4")])

assert VERTEX_DTYPE.itemsize == 52 and FACE_DTYPE.itemsize == 52 and MATERIAL_DTYPE.itemsize == 100
assert LIGHT_DTYPE.itemsize == 44 and BVH_NODE_DTYPE.itemsize == 40 and LST_NODE_DTYPE.itemsize == 8
assert C.sizeof(Camera) == 40


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise OSError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"or `make -C pathtracer_amd/csrc` first")
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_host = None
_core = None


def host_lib():
    """libpthost.so with argument types set."""
    global _host
    if _host is not None:
        return _host
    L = _load("libpthost.so")
    L.pth_last_error.restype = C.c_char_p
    L.pth_scene_load_gltf.argtypes = [C.c_char_p, C.POINTER(HostConfig), C.POINTER(C.c_void_p)]
    L.pth_scene_from_desc.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
    L.pth_scene_get_desc.argtypes = [C.c_void_p]
    L.pth_scene_get_desc.restype = C.POINTER(SceneDesc)
    L.pth_scene_free.argtypes = [C.c_void_p]
    L.pth_scene_free.restype = None
    L.pth_calculate_tangents.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32]
    L.pth_load_hdr.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_float))]
    L.pth_load_png.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint8))]
    L.pth_free.argtypes = [C.c_void_p]
    L.pth_free.restype = None
    L.pth_bvh_build.argtypes = [C.POINTER(SceneDesc), C.POINTER(HostBvh)]
    L.pth_bvh_free.argtypes = [C.POINTER(HostBvh)]
    L.pth_bvh_free.restype = None
    L.pth_lst_build.argtypes = [C.POINTER(SceneDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
    L.pth_tonemap_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.pth_write_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
    L.pth_write_png_rgb8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_char_p]
    L.pth_load_config.argtypes = [C.POINTER(HostConfig), C.c_int, C.POINTER(C.c_char_p)]
    L.pth_log_init.argtypes = [C.c_int, C.c_char_p, C.c_int]
    L.pth_log_close.restype = None
    _host = L
    return L


def core_lib():
    """libptcore.so (HIP kernels) with argument types set. Raises if missing."""
    global _core
    if _core is not None:
        return _core
    # PT_CORE_LIB picks another build of the same ABI (kernel A/B experiments); default is the product
    L = _load(os.environ.get("PT_CORE_LIB", "libptcore.so"))
    L.pt_last_error.restype = C.c_char_p
    L.pt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.pt_destroy.argtypes = [C.c_void_p]
    L.pt_destroy.restype = None
    L.pt_scene_upload.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
    L.pt_bvh_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.pt_lst_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.pt_configure.argtypes = [C.c_void_p, C.POINTER(RenderConfig)]
    L.pt_render_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.pt_render_batch_into.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.pt_render_batch_static.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RenderConfig), C.c_int, C.c_int, C.c_int]
    L.pt_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]
    L.pt_group_destroy.argtypes = [C.c_void_p]
    L.pt_group_destroy.restype = None
    L.pt_group_size.argtypes = [C.c_void_p]
    L.pt_group_context.argtypes = [C.c_void_p, C.c_int]
    L.pt_group_context.restype = C.c_void_p
    L.pt_group_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.pt_read_image.argtypes = [C.c_void_p, C.c_void_p]
    L.pt_host_alloc.argtypes = [C.c_size_t]
    L.pt_host_alloc.restype = C.c_void_p
    L.pt_host_free.argtypes = [C.c_void_p]
    L.pt_host_free.restype = None
    L.pt_clear_image.argtypes = [C.c_void_p]
    L.pt_synchronize.argtypes = [C.c_void_p]
    L.pt_rng_stream.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_int]
    L.pt_primary_hits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.pt_count_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(CoreCounters)]
    L.pt_last_kernel_ms.argtypes = [C.c_void_p]
    L.pt_last_kernel_ms.restype = C.c_float
    L.pt_kernel_ms.argtypes = [C.c_void_p, C.c_int]
    L.pt_kernel_ms.restype = C.c_float
    L.pt_walk_tree.argtypes = [C.c_void_p]
    L.pt_set_arithmetic.argtypes = [C.c_void_p, C.c_int]
    L.pt_arithmetic.argtypes = [C.c_void_p]
    L.pt_group_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(GroupBatch), C.c_int, C.POINTER(C.c_float)]
    L.pt_device_footprint.argtypes = [C.c_void_p]
    L.pt_device_footprint.restype = C.c_uint64
    _core = L
    return L
