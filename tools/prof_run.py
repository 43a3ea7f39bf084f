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
a = ap.parse_args()

gen = {"many_lights": scenes.write_many_lights_standin, "dragon": scenes.write_dragon_standin,
       "cornell": scenes.write_cornell_like, "glass": scenes.write_glass_mirror_standin}[a.scene]
with tempfile.TemporaryDirectory() as d:
    kw = {}
    if a.scene == "glass":
        kw["clear_texture"] = scenes.write_sky_hdr(d + "/sky.hdr")
    scene = host.Scene.load_gltf(gen(d), **kw)
bvh, lst = host.BVH(scene), host.LST(scene)
r = core.Renderer(0)
r.upload(scene, bvh, lst)
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
        # s_memtime ticks at 100 MHz (constant clock): wave-seconds resident / (slots * kernel time)
        print(f"instrumented launch {dt*1e3:.1f} ms; wave-seconds {c['wave_ticks']/1e8:.2f}; "
              f"mean resident waves {c['wave_ticks']/1e8/dt:.0f}")
r.close()
