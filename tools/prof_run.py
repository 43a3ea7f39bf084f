#!/usr/bin/env python3
"""A few batches of one BASELINE configuration for profiling (rocprofv3 --pmc / --kernel-trace).

  python3 tools/prof_run.py [--config c2] [--arithmetic exact|fast] [--spp N] [--batches 2] [--count]
The scene, size and default samples per batch are bench.py's (CONFIGS); the LAST render launch is the one
tools/summarize_profiles.py reads (the launches before it time the tree forms and record the tile costs).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pathtracer_amd import core  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c2", choices=sorted(bench.CONFIGS))
ap.add_argument("--arithmetic", default="exact", choices=("exact", "fast"))
ap.add_argument("--spp", type=int, default=None)
ap.add_argument("--res", type=int, default=None)
ap.add_argument("--batches", type=int, default=2)
ap.add_argument("--count", action="store_true")
a = ap.parse_args()

cfg = bench.CONFIGS[a.config]
res, spp = a.res or cfg["res"], a.spp or cfg["spp"]
t0 = time.perf_counter()
scene, bvh, lst = bench.load_config_scene(a.config)
print(f"host: scene + BVH + LST {time.perf_counter() - t0:.1f} s; faces {len(scene.faces)}, bvh nodes {bvh.node_count}, "
      f"stats {bvh.stats}, lst {len(lst.nodes)}", flush=True)
if os.environ.get("PT_HOST_ONLY"):
    sys.exit(0)
t0 = time.perf_counter()
r = core.Renderer(0)
r.set_arithmetic(a.arithmetic)
r.upload(scene, bvh, lst)
print(f"upload {time.perf_counter() - t0:.1f} s, device footprint {r.device_footprint() / 2**20:.0f} MiB", flush=True)
r.configure(res, res, 0.0, cfg["clear"])
for b in range(a.batches):
    r.render_batch(42 + b, b * spp, spp)
    ms = r.last_kernel_ms()
    print(f"batch {b}: {ms:.3f} ms  {res * res * spp / ms / 1e3:.2f} Msamples/s  tree {r.walk_tree()} {a.arithmetic}", flush=True)
if a.count:
    t0 = time.perf_counter()
    c = r.count_batch(42, spp)
    dt = time.perf_counter() - t0
    print(c)
    tree = r.walk_tree()
    print("algorithmic_bytes", bench.algorithmic_bytes(c, res, res, tree))
    if "wave_ticks" in c:
        # wave_ticks are shader-clock ticks summed over waves: / launch time = resident waves * clock
        print(f"instrumented launch {dt*1e3:.1f} ms; resident waves x clock = {c['wave_ticks']/dt/1e9:.0f} G/s")
r.close()
