"""Regenerates the fixtures under tests/golden/ (run here, where /root/reference is mounted):

  cornell_scene.npz   the reference's cornell.gltf after OUR host pipeline (transformed vertices
                      with tangents, faces, materials, camera) — derived arrays, not the asset —
                      so BASELINE config 1 can be rendered on the GPU box, which has no reference.
  dragon_scene.npz, glass_mirror_scene.npz
                      the same for dragon.gltf (BASELINE config 3) and glass_mirror.gltf (config 4, with its
                      point light); dragon is 69 747 vertices / 139 482 faces, 10.9 MB of records before
                      compression (the vertex and face records are stored as they cross the C ABI).
  rose_garden_env.npz config 4's environment map: sunny_rose_garden_1k.hdr decoded by OUR .hdr reader
                      (bit-equal to the reference's stb_image, tests/test_reference_pins.py) and point-sampled
                      down to 256x128 RGB float texels (every 4th texel of every 4th row; alpha is 1).
  oracle_c1.json      digests of what the ORACLE produces for config 1 (256x256, 16 spp, seed 42):
                      sha256 of the linear image, of the tonemapped RGB8 buffer, of the jittered
                      primary hit ids, plus integer counters. These pin the oracle against
                      accidental change; they are not reference outputs (parity unpinned, see
                      oracle/oracle.h).
  scene_stats.json    structure counts of the bundled scenes after the host pipeline, next to the
                      figures SURVEY.md §3.3/§8 recorded from the reference's own builder.

Usage: python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from pathtracer_amd import host  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ASSETS = "/root/reference/assets"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def save_scene(name):
    sc = host.Scene.load_gltf(f"{ASSETS}/{name}.gltf")
    cam = sc.camera
    camera = np.array(list(cam["position"]) + list(cam["target"]) + list(cam["updir"]) + [cam["yfov"]], np.float32)
    np.savez_compressed(os.path.join(HERE, f"{name}_scene.npz"),
                        vertices=np.frombuffer(sc.vertices.tobytes(), np.uint8),
                        faces=np.frombuffer(sc.faces.tobytes(), np.uint8),
                        materials=np.frombuffer(sc.materials.tobytes(), np.uint8),
                        lights=np.frombuffer(sc.lights.tobytes(), np.uint8), camera=camera)
    return sc


def main():
    sc = save_scene("cornell")
    save_scene("dragon")
    save_scene("glass_mirror")
    env = host.load_hdr(f"{ASSETS}/sunny_rose_garden_1k.hdr")
    assert env.shape == (512, 1024, 4) and (env[..., 3] == 1.0).all()
    np.savez_compressed(os.path.join(HERE, "rose_garden_env.npz"), rgb=np.ascontiguousarray(env[::4, ::4, :3]))
    bvh, lst = host.BVH(sc), host.LST(sc)
    o = ob.Oracle(sc, bvh, lst, 256, 256)
    img, ctr = o.render_batch(42, 0, 16)
    ids, _ = o.primary_hits(42, True)
    rgb8 = host.tonemap_rgb8(img)
    json.dump({"config": "cornell.gltf 256x256 16spp seed 42 exposure 0 clear 0 0 0",
               "image_sha256": sha(img), "rgb8_sha256": sha(rgb8), "primary_ids_sha256": sha(ids),
               "image_mean": [float(v) for v in img.mean(axis=(0, 1))], "counters": ctr,
               "rng_first8_seed42_sub0": [int(v) for v in ob.rng_stream(42, 0, 8)],
               "rng_first8_seed42_sub65535": [int(v) for v in ob.rng_stream(42, 65535, 8)],
               "rng_first8_seed43_sub1": [int(v) for v in ob.rng_stream(43, 1, 8)]},
              open(os.path.join(HERE, "oracle_c1.json"), "w"), indent=1)

    stats = {}
    survey = {"cornell": dict(vertices=2549, faces=1940, materials=6, node_count=2215, max_depth=16),
              "glass_mirror": dict(vertices=1142, faces=1932, materials=6, node_count=2027, max_depth=14),
              "dragon": dict(vertices=69747, faces=139482, materials=3, node_count=168503, max_depth=25)}
    for name in survey:
        s = host.Scene.load_gltf(f"{ASSETS}/{name}.gltf")
        b = host.BVH(s)
        l = host.LST(s)
        stats[name] = {"ours": dict(vertices=len(s.vertices), faces=len(s.faces), materials=len(s.materials),
                                    lights=len(s.lights), lst_nodes=len(l.nodes), **b.stats,
                                    vertices_sha256=sha(s.vertices), faces_sha256=sha(s.faces),
                                    nodes_sha256=sha(b.nodes), indices_sha256=sha(b.indices)),
                       "survey": survey[name]}
    json.dump(stats, open(os.path.join(HERE, "scene_stats.json"), "w"), indent=1)
    print("golden fixtures written")


if __name__ == "__main__":
    main()
