"""Procedural stand-in scenes, written as glTF + .bin so they pass through the real loader.

The GPU box has no copy of the reference's assets, and four of them (many_lights,
bust, single_light, two_lights) ship without their .bin (reference
.MISSING_LARGE_BLOBS). Each generator below builds a scene with the same STRUCTURE as
the configuration it stands in for: the same room (a unit cube open towards +x; the
numbers are just cube corners), the same camera / light / emitter placement (taken from
the .gltf JSON, which does ship), the same materials, and a mesh with the same triangle
count and local bounds as the missing scan. Everything is deterministic; nothing is read
from the reference tree. Results on these scenes are always labelled "stand-in".
"""
import json
import os
import struct

import numpy as np

F32 = 5126
U16 = 5123
U32 = 5125


class GltfWriter:
    """Minimal glTF 2.0 writer: float VEC2/3/4 attributes, u16/u32 indices, TRS nodes."""

    def __init__(self):
        self.bin = bytearray()
        self.doc = {"asset": {"generator": "pathtracer_amd.scenes", "version": "2.0"}, "scene": 0,
                    "scenes": [{"name": "Scene", "nodes": []}], "nodes": [], "materials": [], "meshes": [],
                    "accessors": [], "bufferViews": [], "buffers": []}

    def _view(self, data, target):
        while len(self.bin) % 4:
            self.bin.append(0)
        off = len(self.bin)
        self.bin += data
        self.doc["bufferViews"].append({"buffer": 0, "byteLength": len(data), "byteOffset": off, "target": target})
        return len(self.doc["bufferViews"]) - 1

    def accessor(self, arr, kind):
        arr = np.ascontiguousarray(arr)
        if kind == "SCALAR":
            comp = U16 if arr.dtype == np.uint16 else U32
            view = self._view(arr.tobytes(), 34963)
            acc = {"bufferView": view, "componentType": comp, "count": int(arr.size), "type": "SCALAR"}
        else:
            arr = arr.astype(np.float32)
            view = self._view(arr.tobytes(), 34962)
            acc = {"bufferView": view, "componentType": F32, "count": int(arr.shape[0]), "type": kind}
            if kind == "VEC3":
                acc["min"] = [float(v) for v in arr.min(axis=0)]
                acc["max"] = [float(v) for v in arr.max(axis=0)]
        self.doc["accessors"].append(acc)
        return len(self.doc["accessors"]) - 1

    def texture(self, uri, wrap_s=10497, wrap_t=10497, mag_filter=9729):
        """Image file (relative uri) + its own sampler; returns the texture index."""
        self.doc.setdefault("images", []).append({"uri": uri})
        self.doc.setdefault("samplers", []).append({"magFilter": mag_filter, "minFilter": 9987, "wrapS": wrap_s, "wrapT": wrap_t})
        self.doc.setdefault("textures", []).append({"source": len(self.doc["images"]) - 1,
                                                    "sampler": len(self.doc["samplers"]) - 1})
        return len(self.doc["textures"]) - 1

    def material(self, name, base_color, metallic=0.0, roughness=0.5, emissive=None, emissive_strength=None,
                 extensions=None, omit_metallic=False, base_color_texture=None, metallic_roughness_texture=None,
                 alpha_mode=None, alpha_cutoff=None):
        pbr = {"baseColorFactor": [float(c) for c in base_color], "roughnessFactor": float(roughness)}
        if not omit_metallic:
            pbr["metallicFactor"] = float(metallic)
        if base_color_texture is not None:
            pbr["baseColorTexture"] = {"index": base_color_texture}
        if metallic_roughness_texture is not None:
            pbr["metallicRoughnessTexture"] = {"index": metallic_roughness_texture}
        m = {"doubleSided": True, "name": name, "pbrMetallicRoughness": pbr}
        if alpha_mode is not None:
            m["alphaMode"] = alpha_mode
        if alpha_cutoff is not None:
            m["alphaCutoff"] = float(alpha_cutoff)
        ext = dict(extensions or {})
        if emissive is not None:
            m["emissiveFactor"] = [float(c) for c in emissive]
            ext["KHR_materials_emissive_strength"] = {"emissiveStrength": float(emissive_strength)}
        if ext:
            m["extensions"] = ext
        self.doc["materials"].append(m)
        return len(self.doc["materials"]) - 1

    def primitive(self, positions, normals, indices, material, uvs=None, tangents=None):
        n = len(positions)
        idx = np.asarray(indices).reshape(-1)
        idx = idx.astype(np.uint16) if n <= 65535 else idx.astype(np.uint32)
        attrs = {"POSITION": self.accessor(positions, "VEC3"), "NORMAL": self.accessor(normals, "VEC3")}
        if uvs is not None:
            attrs["TEXCOORD_0"] = self.accessor(uvs, "VEC2")
        if tangents is not None:
            attrs["TANGENT"] = self.accessor(tangents, "VEC4")
        return {"attributes": attrs, "indices": self.accessor(idx, "SCALAR"), "material": material}

    def mesh(self, name, primitives):
        self.doc["meshes"].append({"name": name, "primitives": primitives})
        return len(self.doc["meshes"]) - 1

    def node(self, name, mesh=None, camera=None, light=None, translation=None, rotation=None, scale=None):
        n = {"name": name}
        if mesh is not None:
            n["mesh"] = mesh
        if camera is not None:
            n["camera"] = camera
        if light is not None:
            n["extensions"] = {"KHR_lights_punctual": {"light": light}}
        if translation is not None:
            n["translation"] = [float(v) for v in translation]
        if rotation is not None:
            n["rotation"] = [float(v) for v in rotation]
        if scale is not None:
            n["scale"] = [float(v) for v in scale]
        self.doc["nodes"].append(n)
        self.doc["scenes"][0]["nodes"].append(len(self.doc["nodes"]) - 1)

    def camera(self, yfov, aspect=1.0):
        self.doc.setdefault("cameras", []).append(
            {"name": "Camera", "type": "perspective",
             "perspective": {"aspectRatio": float(aspect), "yfov": float(yfov), "zfar": 1000, "znear": 0.1}})
        return len(self.doc["cameras"]) - 1

    def light(self, kind, intensity, color=(1, 1, 1)):
        ext = self.doc.setdefault("extensions", {}).setdefault("KHR_lights_punctual", {"lights": []})
        ext["lights"].append({"color": [float(c) for c in color], "intensity": float(intensity), "type": kind, "name": kind})
        used = self.doc.setdefault("extensionsUsed", [])
        if "KHR_lights_punctual" not in used:
            used.append("KHR_lights_punctual")
        return len(ext["lights"]) - 1

    def write(self, directory, name):
        os.makedirs(directory, exist_ok=True)
        self.doc["buffers"] = [{"byteLength": len(self.bin), "uri": name + ".bin"}]
        with open(os.path.join(directory, name + ".bin"), "wb") as f:
            f.write(self.bin)
        path = os.path.join(directory, name + ".gltf")
        with open(path, "w") as f:
            json.dump(self.doc, f)
        return path


# ---- geometry helpers ---------------------------------------------------------------------

def quad(p00, p10, p01, p11, normal, uv=((0, 1), (1, 1), (0, 0), (1, 0))):
    pos = np.array([p00, p10, p01, p11], np.float32)
    nrm = np.tile(np.array(normal, np.float32), (4, 1))
    return pos, nrm, np.array(uv, np.float32), np.array([0, 1, 3, 0, 3, 2], np.uint32)


def room_primitives(w, white, red, blue):
    """The three wall primitives of the reference's room mesh: back/floor/ceiling, -y wall, +y wall."""
    pos = np.array([[-1, -1, -1], [-1, -1, -1], [-1, -1, 1], [-1, -1, 1], [-1, 1, -1], [-1, 1, -1], [-1, 1, 1], [-1, 1, 1],
                    [1, -1, -1], [1, -1, 1], [1, 1, -1], [1, 1, 1]], np.float32)
    nrm = np.array([[-1, 0, 0], [0, 0, -1], [-1, 0, 0], [0, 0, 1], [-1, 0, 0], [0, 0, -1], [-1, 0, 0], [0, 0, 1],
                    [0, 0, -1], [0, 0, 1], [0, 0, -1], [0, 0, 1]], np.float32)
    uv = np.array([[.375, 1], [.125, .25], [.625, 1], [.875, .25], [.375, .75], [.125, .5], [.625, .75], [.875, .5],
                   [.375, .25], [.625, .25], [.375, .5], [.625, .5]], np.float32)
    idx = np.array([0, 2, 6, 0, 6, 4, 5, 10, 8, 5, 8, 1, 11, 7, 3, 11, 3, 9], np.uint32)
    prims = [w.primitive(pos, nrm, idx, white, uv)]
    p = np.array([[-1, -1, -1], [-1, -1, 1], [1, -1, -1], [1, -1, 1]], np.float32)
    prims.append(w.primitive(p, np.tile(np.float32([0, -1, 0]), (4, 1)), np.array([2, 3, 1, 2, 1, 0]), red,
                             np.float32([[.375, 0], [.625, 0], [.375, .25], [.625, .25]])))
    p = np.array([[-1, 1, -1], [-1, 1, 1], [1, 1, -1], [1, 1, 1]], np.float32)
    prims.append(w.primitive(p, np.tile(np.float32([0, 1, 0]), (4, 1)), np.array([0, 1, 3, 0, 3, 2]), blue,
                             np.float32([[.375, .75], [.625, .75], [.375, .5], [.625, .5]])))
    return prims


def unit_plane(w, material):
    pos = np.float32([[-1, -1, 0], [1, -1, 0], [-1, 1, 0], [1, 1, 0]])
    nrm = np.tile(np.float32([0, 0, 1]), (4, 1))
    uv = np.float32([[0, 1], [1, 1], [0, 0], [1, 0]])
    return w.primitive(pos, nrm, np.array([0, 1, 3, 0, 3, 2]), material, uv)


def blob(rings, segments, center=(0, 0, 0), radii=(1, 1, 1), bump=0.0, seed=1, with_uv=True):
    """Displaced UV-ellipsoid: (rings+1)*(segments+1) vertices, 2*rings*segments triangles."""
    rng = np.random.RandomState(seed)
    th = np.linspace(0.02, np.pi - 0.02, rings + 1)[:, None]
    ph = np.linspace(0.0, 2 * np.pi, segments + 1)[None, :]
    r = np.ones((rings + 1, segments + 1))
    if bump:
        for _ in range(6):
            a, b = rng.randint(1, 9), rng.randint(1, 7)
            p0, p1 = rng.uniform(0, 2 * np.pi, 2)
            r = r + (bump / 6.0) * np.sin(a * th + p0) * np.cos(b * ph + p1)
    x = r * np.sin(th) * np.cos(ph)
    y = r * np.sin(th) * np.sin(ph)
    z = r * np.cos(th) * np.ones_like(ph)
    P = np.stack([x * radii[0] + center[0], y * radii[1] + center[1], z * radii[2] + center[2]], -1)
    # normals from the parametrisation, periodic in phi
    core = P[:, :-1]
    d_ph = np.roll(core, -1, axis=1) - np.roll(core, 1, axis=1)
    d_th = np.gradient(core, axis=0)
    N = np.cross(d_th, d_ph)
    N /= np.maximum(np.linalg.norm(N, axis=-1, keepdims=True), 1e-20)
    N = np.concatenate([N, N[:, :1]], axis=1)
    uv = np.stack([np.broadcast_to(ph / (2 * np.pi), x.shape), np.broadcast_to(th / np.pi, x.shape)], -1)
    s1 = segments + 1
    rr, ss = np.meshgrid(np.arange(rings), np.arange(segments), indexing="ij")
    a = (rr * s1 + ss).ravel()
    b = a + 1
    c = a + s1
    d = c + 1
    idx = np.stack([a, c, b, b, c, d], -1).astype(np.uint32).reshape(-1)
    return (P.reshape(-1, 3).astype(np.float32), N.reshape(-1, 3).astype(np.float32),
            uv.reshape(-1, 2).astype(np.float32) if with_uv else None, idx)


ROOM_CAMERA = dict(rotation=(0.5, 0.5, 0.5, 0.5), translation=(4.107016563415527, 0, 0), yfov=0.6911112070083618)


def _room(w, emit_strength):
    white = w.material("WHITE", (0.8, 0.8, 0.8, 1))
    red = w.material("RED", (1, 0, 0, 1))
    blue = w.material("BLUE", (0, 0, 1, 1))
    w.doc.setdefault("extensionsUsed", []).append("KHR_materials_emissive_strength")
    w.node("Cube", mesh=w.mesh("Cube.001", room_primitives(w, white, red, blue)))
    w.node("Camera", camera=w.camera(ROOM_CAMERA["yfov"]), rotation=ROOM_CAMERA["rotation"],
           translation=ROOM_CAMERA["translation"])
    return white


def write_cornell_like(directory, name="cornell_like", detail=16):
    """Room + ceiling emitter + smooth sphere + bumpy blob (the shape of cornell.gltf)."""
    w = GltfWriter()
    _room(w, 3)
    emit = w.material("EMIT", (0, 0, 0, 1), emissive=(1, 1, 1), emissive_strength=3)
    sphere = w.material("SPHERE1", (0.353, 0.346, 0.342, 1), roughness=0.0)
    monkey = w.material("MONKEY", (0.8, 0.8, 0.8, 1))
    w.node("Plane.001", mesh=w.mesh("Plane.001", [unit_plane(w, emit)]), rotation=(-1, 0, 0, 1.6292069915380125e-07),
           scale=(0.345,) * 3, translation=(0.0116, 8.3e-05, 0.9759))
    p, n, uv, idx = blob(detail, 2 * detail)
    w.node("Sphere", mesh=w.mesh("Sphere", [w.primitive(p, n, idx, sphere, uv)]), scale=(0.3245,) * 3,
           translation=(-0.408, 0.459, -0.687))
    p, n, uv, idx = blob(detail, 2 * detail, bump=0.35, seed=7)
    w.node("Blob", mesh=w.mesh("Blob", [w.primitive(p, n, idx, monkey, uv)]),
           rotation=(-0.0935, -0.3066, 0.7899, 0.5229), scale=(0.3,) * 3, translation=(0.208, -0.413, -0.70))
    return w.write(directory, name)


def write_many_lights_standin(directory, name="many_lights_standin", rings=314, segments=514):
    """Stand-in for many_lights.gltf: room, point light, sun, emitter plane (strength 5) and a
    bumpy ellipsoid with the scan's triangle count (2*314*514 = 322 792) and local bounds."""
    w = GltfWriter()
    _room(w, 5)
    white2 = w.material("WHITE.001", (0.8, 0.8, 0.8, 1))
    emit = w.material("EMIT", (0, 0, 0, 1), emissive=(1, 1, 1), emissive_strength=5)
    lo = np.array([-0.4855581820011139, -0.3106273114681244, -0.07203955948352814])
    hi = np.array([0.3297273516654968, 0.3140838146209717, 1.1006748676300049])
    p, n, uv, idx = blob(rings, segments, center=(lo + hi) / 2, radii=(hi - lo) / 2 / 1.25, bump=0.25, seed=3)
    w.node("pp-exported.002", mesh=w.mesh("pp-exported.002", [w.primitive(p, n, idx, white2, uv)]),
           rotation=(-0.005002109799534082, 0.005505907349288464, 0.6724141240119934, 0.7401377558708191),
           scale=(1.2687541246414185,) * 3, translation=(0, 0, -0.9272785186767578))
    w.node("Point", light=w.light("point", 271.7570653294113),
           translation=(0.29129457473754883, 0.4542418122291565, 0.5518084168434143))
    w.node("Sun", light=w.light("directional", 271),
           rotation=(0.23479531705379486, 0.2545164227485657, 0.3653407096862793, 0.8640710115432739),
           translation=(0, -0.46805620193481445, 0.7157825231552124))
    w.node("Plane", mesh=w.mesh("Plane", [unit_plane(w, emit)]), rotation=(0.7071068286895752, 0, 0, 0.7071068286895752),
           scale=(0.20217204093933105,) * 3, translation=(0, -0.9694505929946899, 0.43223485350608826))
    return w.write(directory, name)


def write_dragon_standin(directory, name="dragon_standin", rings=264, segments=264):
    """Stand-in for dragon.gltf: ground plane, emitter plane and a bumpy blob WITHOUT texture
    coordinates (so, like the dragon, every vertex gets the default tangent (1,0,0,+1))."""
    w = GltfWriter()
    w.doc.setdefault("extensionsUsed", []).append("KHR_materials_emissive_strength")
    ground = w.material("GROUND", (0.8000229597091675, 0.539573609828949, 0.14442956447601318, 1), roughness=0.0)
    dragon = w.material("DRAGON", (0.8, 0.8, 0.8, 1), roughness=0.0)
    emit = w.material("Material", (0.8, 0.8, 0.8, 1), emissive=(1, 1, 1), emissive_strength=3)
    w.node("Plane", mesh=w.mesh("Plane", [unit_plane(w, ground)]), scale=(0.36600080132484436,) * 3)
    w.node("Camera", camera=w.camera(0.36033553335070556, 4 / 3),
           rotation=(0.6299581527709961, -0.06638966500759125, -0.0810980424284935, 0.7695246934890747),
           scale=(0.1901395320892334, 0.1901395469903946, 0.1901395469903946),
           translation=(-0.13736775517463684, -0.6285547018051147, 0.19172675907611847))
    lo = np.array([-0.10834388434886932, -0.04121911898255348, 3.628234117059037e-05])
    hi = np.array([0.09654387831687927, 0.0503932386636734, 0.14443814754486084])
    p, n, _, idx = blob(rings, segments, center=(lo + hi) / 2, radii=(hi - lo) / 2 / 1.3, bump=0.3, seed=11, with_uv=False)
    w.node("Dragon", mesh=w.mesh("Dragon", [w.primitive(p, n, idx, dragon)]))
    w.node("Plane.001", mesh=w.mesh("Plane.001", [unit_plane(w, emit)]), scale=(0.09044785052537918,) * 3,
           translation=(0.0057862140238285065, 0.09616303443908691, 0.26937922835350037))
    return w.write(directory, name)


def write_glass_mirror_standin(directory, name="glass_mirror_standin", detail=24):
    """Stand-in for glass_mirror.gltf: room, point light, emitter plane, a 'mirror' sphere
    (metallicFactor omitted -> 1, roughness 0) and a 'glass' sphere (transmission/ior extensions,
    which the reference parses and never shades with)."""
    w = GltfWriter()
    _room(w, 5)
    emit = w.material("EMIT", (0, 0, 0, 1), emissive=(1, 1, 1), emissive_strength=5)
    mirror = w.material("MIRROR", (1, 0.9382134079933167, 0.9287373423576355, 1), roughness=0.0, omit_metallic=True)
    glass = w.material("GLASS", (0.8, 0.8, 0.8, 1), roughness=0.0,
                       extensions={"KHR_materials_transmission": {"transmissionFactor": 1},
                                   "KHR_materials_ior": {"ior": 1.3329999446868896}})
    w.node("Point", light=w.light("point", 103.26770296691764),
           translation=(-0.07358639687299728, 0.6332712769508362, 0.29365500807762146))
    w.node("Plane", mesh=w.mesh("Plane", [unit_plane(w, emit)]), rotation=(0.7071068286895752, 0, 0, 0.7071067690849304),
           scale=(0.2977488338947296,) * 3, translation=(0, -0.985132098197937, 0.2459760308265686))
    p, n, uv, idx = blob(detail, 2 * detail)
    m = w.mesh("Sphere", [w.primitive(p, n, idx, mirror, uv)])
    w.node("Sphere", mesh=m, scale=(0.30951082706451416,) * 3,
           translation=(-0.18294227123260498, 0.5162081122398376, -0.6951195001602173))
    m2 = w.mesh("Sphere.001", [w.primitive(p, n, idx, glass, uv)])
    w.node("Sphere.001", mesh=m2, scale=(0.30951082706451416,) * 3,
           translation=(0.24995926022529602, -0.2790658473968506, -0.6951195001602173))
    return w.write(directory, name)


def write_png_rgba8(path, img):
    """Plain RGBA8 PNG (filter 0, one zlib stream) from an (H, W, 4) uint8 array."""
    import struct
    import zlib
    img = np.ascontiguousarray(img, np.uint8)
    h, w, _ = img.shape

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
    return path


def write_textured_standin(directory, name="textured_standin", detail=20):
    """Room whose objects exercise the material texture path (scene.cu:388-470, intersect.cu:137-160):
    a checker base-colour texture with an alpha channel on a MASK material, a metallic-roughness
    texture, a BLEND quad whose alpha drives the stochastic pass-through (renderer.cu:310-314), and
    the three wrap modes / both filters (uvs run outside [0,1])."""
    os.makedirs(directory, exist_ok=True)
    rng = np.random.RandomState(3)
    n = 32
    yy, xx = np.mgrid[0:n, 0:n]
    checker = (((xx // 4) + (yy // 4)) % 2).astype(np.uint8)
    base = np.zeros((n, n, 4), np.uint8)
    base[..., 0] = 60 + 180 * checker
    base[..., 1] = 200 - 120 * checker
    base[..., 2] = rng.randint(40, 255, (n, n))
    base[..., 3] = np.where((xx + yy) % 7 == 0, 40, 255)  # holes for the MASK cutoff
    write_png_rgba8(os.path.join(directory, name + "_base.png"), base)
    mr = np.zeros((n, n, 4), np.uint8)
    mr[..., 1] = rng.randint(30, 255, (n, n))   # roughness (G)
    mr[..., 2] = 255 * checker                  # metallic (B)
    mr[..., 3] = 255
    write_png_rgba8(os.path.join(directory, name + "_mr.png"), mr)
    veil = np.zeros((8, 8, 4), np.uint8)
    veil[..., :3] = 230
    veil[..., 3] = np.linspace(30, 220, 64).reshape(8, 8).astype(np.uint8)
    write_png_rgba8(os.path.join(directory, name + "_veil.png"), veil)

    w = GltfWriter()
    _room(w, 4)
    emit = w.material("EMIT", (0, 0, 0, 1), emissive=(1, 1, 1), emissive_strength=4)
    t_base = w.texture(name + "_base.png", wrap_s=33648, wrap_t=10497, mag_filter=9729)   # mirrored / repeat, linear
    t_mr = w.texture(name + "_mr.png", wrap_s=33071, wrap_t=33071, mag_filter=9728)       # clamp, nearest
    t_veil = w.texture(name + "_veil.png")
    cut = w.material("CUTOUT", (1, 1, 1, 1), roughness=1.0, metallic=1.0, base_color_texture=t_base,
                     metallic_roughness_texture=t_mr, alpha_mode="MASK", alpha_cutoff=0.5)
    veil_m = w.material("VEIL", (0.9, 0.9, 1.0, 0.8), roughness=0.4, base_color_texture=t_veil, alpha_mode="BLEND")
    w.node("Plane.001", mesh=w.mesh("Plane.001", [unit_plane(w, emit)]), rotation=(-1, 0, 0, 1.6292069915380125e-07),
           scale=(0.345,) * 3, translation=(0.0116, 8.3e-05, 0.9759))
    p, nrm, uv, idx = blob(detail, 2 * detail, bump=0.2, seed=5)
    w.node("Cutout", mesh=w.mesh("Cutout", [w.primitive(p, nrm, idx, cut, uv * 2.5 - 0.6)]), scale=(0.45,) * 3,
           translation=(-0.2, 0.1, -0.45))
    pos, qn, quv, qi = quad((0.6, -0.7, -0.9), (0.6, 0.7, -0.9), (0.6, -0.7, 0.5), (0.6, 0.7, 0.5), (1, 0, 0))
    w.node("Veil", mesh=w.mesh("Veil", [w.primitive(pos, qn, qi, veil_m, quv)]))
    w.node("Light", light=w.light("point", 40.0), translation=(0.8, -0.5, 0.6))
    return w.write(directory, name)


def write_synthetic_grid(directory, name="synthetic_grid", side=2236, emitters=8):
    """C5: a displaced height field of side*side quads (2 triangles each; side=2236 -> 10.0 M
    triangles, 5.0 M vertices) under `emitters` emissive quads, TANGENT supplied analytically so
    the loader does not need to derive 30 M corner tangents. LST = 2*emitters faces."""
    w = GltfWriter()
    w.doc.setdefault("extensionsUsed", []).append("KHR_materials_emissive_strength")
    ground = w.material("TERRAIN", (0.7, 0.65, 0.55, 1), roughness=0.6)
    emit = w.material("EMIT", (0, 0, 0, 1), emissive=(1, 0.95, 0.9), emissive_strength=40)
    n = side + 1
    u = np.linspace(-1.0, 1.0, n)
    X, Y = np.meshgrid(u, u, indexing="xy")

    def height(x, y):
        return (0.08 * np.sin(9.0 * x + 0.3) * np.cos(7.0 * y - 0.2) + 0.03 * np.sin(31.0 * x * y + 1.0)
                + 0.015 * np.cos(83.0 * x + 1.7) * np.sin(71.0 * y))

    Z = height(X, Y)
    eps = 1e-4
    dzdx = (height(X + eps, Y) - height(X - eps, Y)) / (2 * eps)
    dzdy = (height(X, Y + eps) - height(X, Y - eps)) / (2 * eps)
    P = np.stack([X, Y, Z], -1).reshape(-1, 3).astype(np.float32)
    N = np.stack([-dzdx, -dzdy, np.ones_like(Z)], -1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    T = np.stack([np.ones_like(Z), np.zeros_like(Z), dzdx], -1)
    T /= np.linalg.norm(T, axis=-1, keepdims=True)
    T4 = np.concatenate([T, np.ones_like(Z)[..., None]], -1).reshape(-1, 4).astype(np.float32)
    UV = np.stack([(X + 1) / 2, (Y + 1) / 2], -1).reshape(-1, 2).astype(np.float32)
    ii, jj = np.meshgrid(np.arange(side, dtype=np.uint32), np.arange(side, dtype=np.uint32), indexing="xy")
    a = (jj * n + ii).ravel()
    b = a + 1
    c = a + n
    d = c + 1
    idx = np.stack([a, b, d, a, d, c], -1).reshape(-1)
    w.node("Terrain", mesh=w.mesh("Terrain", [w.primitive(P, N.reshape(-1, 3).astype(np.float32), idx, ground, UV, T4)]))
    prims = []
    rng = np.random.RandomState(1)
    for _ in range(emitters):
        cx, cy = rng.uniform(-0.8, 0.8, 2)
        s = 0.05
        pos = np.float32([[cx - s, cy - s, 0.5], [cx + s, cy - s, 0.5], [cx - s, cy + s, 0.5], [cx + s, cy + s, 0.5]])
        nrm = np.tile(np.float32([0, 0, -1]), (4, 1))
        uv = np.float32([[0, 1], [1, 1], [0, 0], [1, 0]])
        tan = np.tile(np.float32([1, 0, 0, 1]), (4, 1))
        prims.append(w.primitive(pos, nrm, np.array([0, 3, 1, 0, 2, 3]), emit, uv, tan))
    w.node("Emitters", mesh=w.mesh("Emitters", prims))
    # camera above one corner looking at the centre: rotation maps -z to the view direction
    w.node("Camera", camera=w.camera(0.8), translation=(0.0, -1.6, 1.1),
           rotation=(0.5, 0.0, 0.0, 0.8660254037844386))
    return w.write(directory, name)


def write_rgbe(path, img):
    """Radiance .hdr writer (RLE scanlines made of literal runs) for procedural environment maps."""
    img = np.asarray(img, np.float32)
    h, wd, _ = img.shape
    m = img.max(axis=-1)
    e = np.zeros_like(m, dtype=np.int32)
    nz = m > 1e-32
    mant, ex = np.frexp(m[nz])
    scale = np.zeros_like(m)
    scale[nz] = mant * 256.0 / m[nz]
    e[nz] = ex + 128
    rgbe = np.zeros((h, wd, 4), np.uint8)
    rgbe[..., :3] = np.clip(img * scale[..., None], 0, 255).astype(np.uint8)
    rgbe[..., 3] = e.astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(f"-Y {h} +X {wd}\n".encode())
        for y in range(h):
            f.write(struct.pack("BBBB", 2, 2, wd >> 8, wd & 255))
            for ch in range(4):
                row = rgbe[y, :, ch].tobytes()
                for off in range(0, wd, 128):
                    chunk = row[off:off + 128]
                    f.write(bytes([len(chunk)]) + chunk)
    return path


def write_sky_hdr(path, width=256, height=128):
    """A smooth procedural sky with a bright sun patch (stand-in for the bundled 1k HDRs)."""
    v = (np.arange(height) + 0.5) / height
    u = (np.arange(width) + 0.5) / width
    U, V = np.meshgrid(u, v)
    sky = np.stack([0.25 + 0.5 * (1 - V), 0.35 + 0.5 * (1 - V), 0.6 + 0.4 * (1 - V)], -1)
    ground = np.stack([0.25 * np.ones_like(V), 0.22 * np.ones_like(V), 0.18 * np.ones_like(V)], -1)
    img = np.where((V < 0.5)[..., None], sky, ground)
    sun = np.exp(-(((U - 0.3) / 0.02) ** 2 + ((V - 0.25) / 0.03) ** 2))
    img = img + sun[..., None] * np.array([40.0, 36.0, 30.0])
    return write_rgbe(path, img.astype(np.float32))
