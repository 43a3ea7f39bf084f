#!/usr/bin/env python3
"""One row per BASELINE configuration on one MI355X: kernel time per launch (HIP events, median of the launches
after the first), Msamples/s, reference-equivalent Mrays/s, algorithmic GB/s (SURVEY.md section 8d's per-unit bytes
x the instrumented launch's counters) and the CPU oracle's rate on a bounded sample of the same launch.
  python3 tools/config_table.py > profiles/rNN_config_table.md      (on the GPU box)"""
import os
import statistics
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402  (reported CPU baseline only)
from pathtracer_amd import core, scenes  # noqa: E402
from tests import util  # noqa: E402

THREADS = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16"))))


def row(label, b, res, spp, launches, clear=(0.0, 0.0, 0.0), cpu_rows=64):
    r = core.Renderer(0)
    r.upload(b.scene, b.bvh, b.lst)
    r.configure(res, res, 0.0, clear)
    ms = []
    for k in range(launches):
        r.render_batch(42 + k, k * spp, spp)
        ms.append(r.last_kernel_ms())
    t = statistics.median(ms[1:]) if launches > 1 else ms[0]
    c = r.count_batch(42, spp)
    tree = r.walk_tree()
    abytes = (112 if tree == 4 else 64) * c["inner_visits"] + 40 * c["tri_tests"] + 240 * c["shaded_hits"] + 24 * res * res
    foot = r.device_footprint()
    r.close()
    o = ob.Oracle(b.scene, b.bvh, b.lst, res, res, 0.0, clear)
    img = np.zeros((res, res, 3), np.float32)
    t0 = time.perf_counter()
    _, oc = o.render_batch(42, 0, spp, img, threads=THREADS, rows=(res // 2 - cpu_rows // 2, res // 2 + cpu_rows // 2))
    dt = time.perf_counter() - t0
    samples = res * res * spp
    print(f"| {label} | {len(b.scene.faces):,} | {res}² × {spp} | {'4-wide' if tree == 4 else 'binary'} | {t:.1f} | {samples / t / 1e3:.0f} | "
          f"{c['rays'] / t / 1e3:.0f} | {c['rays'] / c['samples']:.2f} | {c['inner_visits'] / c['traversals']:.1f} | {abytes / t / 1e6:.0f} | "
          f"{c['redone_walks']:,} | {foot / 2**20:.0f} | {oc['samples'] / dt / 1e6:.2f} |", flush=True)


def main():
    print("| configuration | faces | launch | tree | ms / launch | Msamples/s | Mrays/s (reference-equivalent) | rays / sample | node visits / walk | "
          "algorithmic GB/s | walks repeated sequentially | device MiB | CPU oracle Msamples/s (%d threads, middle rows) |" % THREADS)
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    with tempfile.TemporaryDirectory() as d:
        row("C1 cornell.gltf (golden arrays)", util.golden_scene("cornell"), 256, 16, 5, cpu_rows=256)
        row("C2 many_lights STAND-IN", util.load_bundle(scenes.write_many_lights_standin(d)), 1024, 50, 5, cpu_rows=128)
        row("C3 dragon.gltf (golden arrays)", util.golden_scene("dragon"), 1024, 64, 5, cpu_rows=128)
        row("C4 glass_mirror.gltf + rose-garden env (golden arrays)", util.golden_scene("glass_mirror", env=True), 1024, 50, 5, clear=(1.0, 1.0, 1.0), cpu_rows=128)
        row("C5 synthetic 10 M-triangle grid, one GPU's share of a batch", util.load_bundle(scenes.write_synthetic_grid(d, side=2236)), 4096, 16, 3, cpu_rows=64)


if __name__ == "__main__":
    main()
