#!/usr/bin/env python3
"""One batch of the radiance kernel for profiling (rocprofv3 --pmc / --kernel-trace).

  python3 tools/prof_run.py [--scene many_lights|dragon|cornell|glass] [--res 1024] [--spp 10] [--batches 1]
"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import core, host, scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="many_lights")
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--spp", type=int, default=10)
ap.add_argument("--batches", type=int, default=1)
ap.add_argument("--count", action="store_true")
ap.add_argument("--side", type=int, default=2236, help="grid scene: quads per side")
a = ap.parse_args()

gen = {"many_lights": scenes.write_many_lights_standin, "dragon": scenes.write_dragon_standin,
       "cornell": scenes.write_cornell_like, "glass": scenes.write_glass_mirror_standin,
       "grid": lambda d: scenes.write_synthetic_grid(d, side=a.side)}[a.scene]
import time
t_host = time.perf_counter()
with tempfile.TemporaryDirectory() as d:
    kw = {}
    if a.scene == "glass":
        kw["clear_texture"] = scenes.write_sky_hdr(d + "/sky.hdr")
    scene = host.Scene.load_gltf(gen(d), **kw)
t_load = time.perf_counter()
bvh, lst = host.BVH(scene), host.LST(scene)
t_bvh = time.perf_counter()
print(f"host: write+load {t_load - t_host:.1f} s, BVH+LST {t_bvh - t_load:.1f} s; faces {len(scene.faces)}, "
      f"bvh nodes {bvh.node_count}, stats {bvh.stats}, lst {len(lst.nodes)}", flush=True)
if os.environ.get("PT_HOST_ONLY"):
    sys.exit(0)
r = core.Renderer(0)
r.upload(scene, bvh, lst)
print(f"upload {time.perf_counter() - t_bvh:.1f} s, device footprint {r.device_footprint() / 2**20:.0f} MiB", flush=True)
r.configure(a.res, a.res, 0.0, (1.0, 1.0, 1.0) if a.scene == "glass" else (0.0, 0.0, 0.0))
for b in range(a.batches):
    r.render_batch(42 + b, b * a.spp, a.spp)
    ms = r.last_kernel_ms()
    print(f"batch {b}: {ms:.3f} ms  {a.res * a.res * a.spp / ms / 1e3:.2f} Msamples/s", flush=True)
if a.count:
    import time
    t0 = time.perf_counter()
    c = r.count_batch(42, a.spp)
    dt = time.perf_counter() - t0
    print(c)
    if "wave_ticks" in c:
        # wave_ticks are shader-clock ticks summed over waves: / launch time = resident waves * clock
        print(f"instrumented launch {dt*1e3:.1f} ms; resident waves x clock = {c['wave_ticks']/dt/1e9:.0f} G/s "
              f"(3072 slots x ~2.4 GHz = 7373 when every slot is busy)")
r.close()
