#!/usr/bin/env python3
"""The same batches on the binary and on the four-wide tree: floats of the two images that differ.
(The binary walk keeps the reference's order of leaves; the four-wide one may resolve an exact distance tie
between two different faces the other way. This counts how often that shows at full size.)

  python3 tools/tree_compare.py [--scene many_lights|dragon|glass|cornell] [--res 1024] [--spp 50] [--batches 2]
"""
import argparse
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import core, host, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="many_lights")
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--spp", type=int, default=50)
ap.add_argument("--batches", type=int, default=2)
a = ap.parse_args()
gen = {"many_lights": scenes.write_many_lights_standin, "dragon": scenes.write_dragon_standin,
       "cornell": scenes.write_cornell_like, "glass": scenes.write_glass_mirror_standin}[a.scene]
with tempfile.TemporaryDirectory() as d:
    kw = {"clear_texture": scenes.write_sky_hdr(d + "/sky.hdr")} if a.scene == "glass" else {}
    scene = host.Scene.load_gltf(gen(d), **kw)
bvh, lst = host.BVH(scene), host.LST(scene)
imgs = {}
for tree in ("binary", "wide"):
    os.environ["PT_TREE"] = tree
    r = core.Renderer(0)
    r.upload(scene, bvh, lst)
    r.configure(a.res, a.res, 0.0, (1.0, 1.0, 1.0) if a.scene == "glass" else (0.0, 0.0, 0.0))
    imgs[tree] = r.render(42, a.spp * a.batches, a.spp)
    print(tree, "walk_tree", r.walk_tree(), "last launch", round(r.last_kernel_ms(), 2), "ms", flush=True)
    r.close()
b, w = imgs["binary"], imgs["wide"]
diff = b.view(np.uint32) != w.view(np.uint32)  # bits, so that NaN pixels (degenerate shading frames) compare equal
both = np.isfinite(b) & np.isfinite(w)
print(f"{a.scene} {a.res}x{a.res} {a.spp * a.batches} spp: {int(diff.sum())} of {diff.size} floats differ in bits, "
      f"{int(diff.any(axis=2).sum())} pixels; non-finite floats {int((~np.isfinite(b)).sum())} / {int((~np.isfinite(w)).sum())}; "
      f"max abs difference where both are finite {float(np.abs(np.where(both, b - w, 0)).max()):.3e}")
ys, xs = np.nonzero(diff.any(axis=2))
for y, x in list(zip(ys, xs))[:8]:
    print(f"  pixel ({x}, {y}): binary {b[y, x]}  wide {w[y, x]}")
