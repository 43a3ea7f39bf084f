"""Pins of the host pipeline against the reference's OWN third-party code, compiled as it is.

oracle/_ref/libgltfref.so = oracle/gltf_oracle.cpp (our restatement of Scene::read_gltf's glue, scene.cu:24-626,
713-972) built against the reference's vendored tinygltf + json.hpp, stb_image and MikkTSpace from where they lie
under /root/reference/src. Everything those libraries decide is therefore the reference's: accessor bytes and
strides, buffers, node TRS values, material defaults, KHR extensions, image texels, tangents.
pathtracer_amd/csrc/host (own JSON reader, own inflate/PNG/RGBE codecs, own tangents) must give the same bytes.

Runs where /root/reference is mounted (this container); skips on the GPU box, where the derived fixtures of
tests/golden/ (made from the loader these tests pin) stand in.
"""
import ctypes as C
import os

import numpy as np
import pytest

from pathtracer_amd import _ffi, host
from tests import util

REF = "/root/reference"
ASSETS = os.path.join(REF, "assets")
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libgltfref.so")

pytestmark = pytest.mark.skipif(not (os.path.isdir(ASSETS) and os.path.exists(LIB)),
                                reason="needs /root/reference and oracle/_ref/libgltfref.so (make -C oracle)")


@pytest.fixture(scope="module")
def gref():
    L = C.CDLL(LIB)
    L.gref_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(_ffi.Camera)]
    L.gref_load.restype = C.c_void_p
    L.gref_error.argtypes = [C.c_void_p]
    L.gref_error.restype = C.c_char_p
    L.gref_desc.argtypes = [C.c_void_p]
    L.gref_desc.restype = C.POINTER(_ffi.SceneDesc)
    L.gref_free.argtypes = [C.c_void_p]
    L.gref_free.restype = None
    for f in (L.gref_stbi_loadf, L.gref_stbi_load):
        f.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        f.restype = C.c_void_p
    L.gref_stbi_free.argtypes = [C.c_void_p]
    L.gref_stbi_free.restype = None
    return L


def _arrays(desc):
    def view(ptr, n, dt):
        return host._view(ptr, n, dt).copy()
    tex = []
    for i in range(desc.texture_count):
        t = desc.textures[i]
        dt = np.dtype("<f4") if t.isFloat else np.dtype("u1")
        tex.append(((t.width, t.height, t.channels, t.isFloat, t.wrapS, t.wrapT, t.filter),
                    host._view(t.pixels, t.width * t.height * 4, dt).copy()))
    c = desc.camera
    cam = np.array([c.position.x, c.position.y, c.position.z, c.target.x, c.target.y, c.target.z,
                    c.updir.x, c.updir.y, c.updir.z, c.yfov], np.float32)
    return {"vertices": view(desc.vertices, desc.vertex_count, _ffi.VERTEX_DTYPE),
            "faces": view(desc.faces, desc.face_count, _ffi.FACE_DTYPE),
            "materials": view(desc.materials, desc.material_count, _ffi.MATERIAL_DTYPE),
            "lights": view(desc.lights, desc.light_count, _ffi.LIGHT_DTYPE),
            "camera": cam, "clearTexture": int(desc.clearTexture), "textures": tex}


CASES = [("cornell", None), ("dragon", None), ("glass_mirror", "sunny_rose_garden_1k.hdr"),
         ("textured_plane", "uv_test.png")]  # (the last: a base-colour texture from the file, a PNG as clear texture)


@pytest.mark.parametrize("name,env", CASES)
def test_loader_output_equals_the_reference_libraries(gref, name, env):
    """Every record the loader hands to the kernel, byte for byte: transformed vertices (positions, normals,
    uvs, tangents incl. MikkTSpace-generated ones), faces, materials (defaults and KHR extensions), lights,
    camera, texture descriptors and texels."""
    path = os.path.join(ASSETS, name + ".gltf")
    envp = os.path.join(ASSETS, env) if env else None
    default_cam = {"position": (0.0, -3.0, 1.0), "target": (0.0, 0.0, 0.5), "updir": (0.0, 0.0, 1.0), "yfov": 0.7}
    ours = host.Scene.load_gltf(path, clear_texture=envp, default_camera=default_cam)
    cam = _ffi.Camera(_ffi.Vec3(*default_cam["position"]), _ffi.Vec3(*default_cam["target"]), _ffi.Vec3(*default_cam["updir"]),
                      default_cam["yfov"])
    h = gref.gref_load(path.encode(), envp.encode() if envp else None, C.byref(cam))
    try:
        assert gref.gref_error(h) == b"", gref.gref_error(h)
        a, b = _arrays(ours.desc), _arrays(gref.gref_desc(h).contents)
    finally:
        gref.gref_free(h)
    assert len(a["vertices"]) == len(b["vertices"]) > 0 and len(a["faces"]) == len(b["faces"]) > 0
    for field in ("position", "normal", "texcoord0", "tangent"):
        assert np.array_equal(util.bits(a["vertices"][field]), util.bits(b["vertices"][field])), field
    for key in ("vertices", "faces", "materials", "lights"):
        assert a[key].tobytes() == b[key].tobytes(), key
    assert np.array_equal(util.bits(a["camera"]), util.bits(b["camera"]))
    assert a["clearTexture"] == b["clearTexture"]
    assert len(a["textures"]) == len(b["textures"])
    for (da, pa), (db, pb) in zip(a["textures"], b["textures"]):
        assert da == db
        assert pa.tobytes() == pb.tobytes()
    if name == "textured_plane":  # its base-colour texture is farmland_overcast_1k.hdr (float), then the PNG (8-bit)
        assert [d[:4] for d, _ in a["textures"]] == [(1024, 512, 4, 1), (1024, 1024, 4, 0)]
        assert a["materials"]["baseColorTexture"][0] == 0
    if name == "glass_mirror":
        assert len(a["lights"]) == 1 and a["clearTexture"] == 0
        # config 4's environment map reaches the kernel at its FULL 1024x512 float resolution, linear filter, mirrored
        # repeat (scene.cu:928-936); the GPU gate's rose_garden_env.npz fixture is these texels point-sampled 4x down
        (w, h, ch, is_float, wrap_s, wrap_t, filt), texels = a["textures"][0]
        assert (w, h, ch, is_float, wrap_s, wrap_t, filt) == (1024, 512, 4, 1, 2, 2, 1) and len(texels) * texels.itemsize == 1024 * 512 * 16


@pytest.mark.parametrize("name", ["sunny_rose_garden_1k.hdr", "farmland_overcast_1k.hdr"])
def test_hdr_decoder_equals_stbi_loadf(gref, name):
    """pth_load_hdr == the reference's stb_image v2.28 stbi_loadf(…, 4), every float of both bundled maps."""
    path = os.path.join(ASSETS, name)
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    p = gref.gref_stbi_loadf(path.encode(), C.byref(w), C.byref(h), C.byref(c))
    assert p and (w.value, h.value, c.value) == (1024, 512, 3)
    want = np.frombuffer((C.c_float * (w.value * h.value * 4)).from_address(p), np.float32).copy()
    gref.gref_stbi_free(p)
    got = host.load_hdr(path)
    assert got.shape == (512, 1024, 4)
    assert np.array_equal(util.bits(got.reshape(-1)), util.bits(want))


def test_png_decoder_equals_stbi_load(gref):
    path = os.path.join(ASSETS, "uv_test.png")
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    p = gref.gref_stbi_load(path.encode(), C.byref(w), C.byref(h), C.byref(c))
    assert p and (w.value, h.value) == (1024, 1024)
    want = np.frombuffer((C.c_uint8 * (w.value * h.value * 4)).from_address(p), np.uint8).copy()
    gref.gref_stbi_free(p)
    got = host.load_png(path)
    assert got.shape == (1024, 1024, 4) and np.array_equal(got.reshape(-1), want)


def test_png_writer_output_decodes_in_stb_image(gref, tmp_path):
    """render.png as the CLI writes it (pth_write_image: tonemap + own deflate / PNG encoder), read back by the
    reference's stb_image: the pixels are the tonemapped buffer, byte for byte (the reference writes with
    stb_image_write, whose exact file bytes depend on its deflate heuristics and are not the contract)."""
    from oracle import binding as ob
    rng = np.random.RandomState(5)
    img = (rng.rand(67, 129, 3).astype(np.float32) ** 3) * 4.0
    path = str(tmp_path / "render.png")
    host.write_image(img, path)
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    p = gref.gref_stbi_load(path.encode(), C.byref(w), C.byref(h), C.byref(c))
    assert p and (w.value, h.value, c.value) == (129, 67, 3)
    got = np.frombuffer((C.c_uint8 * (w.value * h.value * 4)).from_address(p), np.uint8).reshape(67, 129, 4).copy()
    gref.gref_stbi_free(p)
    assert (got[..., 3] == 255).all()
    assert np.array_equal(got[..., :3], ob.tonemap_rgb8(img))
