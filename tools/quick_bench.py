#!/usr/bin/env python3
"""Kernel A/B: ms per 1024x1024x50spp launch of the bench scene (and optionally others) for the library named by
PT_CORE_LIB and the kernel named by PT_KERNEL. Prints one line per scene.
  python3 tools/quick_bench.py [--scenes many_lights,dragon] [--launches 4] [--count]"""
import argparse, os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import core, host, scenes
from tests import util

ap = argparse.ArgumentParser()
ap.add_argument("--scenes", default="many_lights")
ap.add_argument("--launches", type=int, default=4)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--spp", type=int, default=50)
ap.add_argument("--count", action="store_true")
a = ap.parse_args()
tag = f"{os.environ.get('PT_CORE_LIB', 'libptcore.so')} kernel={os.environ.get('PT_KERNEL', 'mega')} stream={os.environ.get('PT_STREAM', '0')} lpt={os.environ.get('PT_LPT', '1')}"
for name in a.scenes.split(","):
    with tempfile.TemporaryDirectory() as d:
        if name == "many_lights":
            b = util.load_bundle(scenes.write_many_lights_standin(d))
        elif name == "dragon":
            b = util.golden_scene("dragon")
        elif name == "glass":
            b = util.golden_scene("glass_mirror", env=True)
        elif name == "grid":
            b = util.load_bundle(scenes.write_synthetic_grid(d, side=2236))
        else:
            raise SystemExit(name)
    r = core.Renderer(0)
    r.upload(b.scene, b.bvh, b.lst)
    r.configure(a.res, a.res, 0.0, (1.0, 1.0, 1.0) if name == "glass" else (0.0, 0.0, 0.0))
    ms = []
    for k in range(a.launches):
        r.render_batch(42 + k, k * a.spp, a.spp)
        ms.append(r.last_kernel_ms())
    line = f"{tag} {name} {a.res}^2x{a.spp}: tree {r.walk_tree()} ms " + " ".join(f"{m:.2f}" for m in ms) + f"  best {min(ms[1:]):.2f}"
    if a.count:
        c = r.count_batch(42, a.spp)
        line += f"\n   lanes {c['lane_utilisation']} time {c.get('time_share')} node {c.get('node_step_lanes')} iters {c['wave_iterations']}"
    print(line, flush=True)
    r.close()
