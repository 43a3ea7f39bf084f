#!/usr/bin/env python3
"""One row per BASELINE configuration on one MI355X: kernel time per launch in exact and in tolerance arithmetic (HIP
events, median of the launches after the first), Msamples/s, reference-equivalent Mrays/s, algorithmic GB/s (SURVEY.md
section 8d's per-unit bytes x the instrumented launch's counters), ISA-counted useful fp32 TFLOP/s, and the CPU
oracle's rate on a bounded sample of the same launch.
  python3 tools/config_table.py > profiles/rNN_config_table.md      (on the GPU box)"""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import binding as ob  # noqa: E402  (reported CPU baseline only)
from pathtracer_amd import core  # noqa: E402

THREADS = max(1, min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16"))))
LABEL = {"c1": "C1 cornell.gltf (golden arrays)", "c2": "C2 many_lights STAND-IN", "c3": "C3 dragon.gltf (golden arrays)",
         "c4": "C4 glass_mirror.gltf + rose-garden env 256x128 (golden arrays)", "c5": "C5 synthetic 10 M-triangle grid, one GPU's share of a batch"}


def measure(r, res, spp, launches):
    ms = []
    for k in range(launches):
        r.render_batch(42 + k, k * spp, spp)
        ms.append(r.last_kernel_ms())
    return statistics.median(ms[1:]) if launches > 1 else ms[0]


def row(name, spp, launches, cpu_rows):
    cfg = bench.CONFIGS[name]
    res, clear = cfg["res"], cfg["clear"]
    scene, bvh, lst = bench.load_config_scene(name)
    out = {}
    for arith in ("exact", "fast"):
        r = core.Renderer(0)
        r.set_arithmetic(arith)
        r.upload(scene, bvh, lst)
        r.configure(res, res, 0.0, clear)
        t = measure(r, res, spp, launches)
        c = r.count_batch(42, spp)
        tree = r.walk_tree()
        isa, _ = bench.newest_profile(f"r*_isa_flops_{arith}*.json")
        flops, _ = bench.algorithmic_flops(c, tree, isa)
        out[arith] = (t, c, tree, bench.algorithmic_bytes(c, res, res, tree), flops, r.device_footprint())
        r.close()
    o = ob.Oracle(scene, bvh, lst, res, res, 0.0, clear)
    img = np.zeros((res, res, 3), np.float32)
    t0 = time.perf_counter()
    _, oc = o.render_batch(42, 0, spp, img, threads=THREADS, rows=(res // 2 - cpu_rows // 2, res // 2 + cpu_rows // 2))
    dt = time.perf_counter() - t0
    samples = res * res * spp
    t, c, tree, abytes, flops, foot = out["exact"]
    tf, cf, treef, _, flopsf, _ = out["fast"]
    tn = {2: "binary", 4: "4-wide"}
    print(f"| {LABEL[name]} | {len(scene.faces):,} | {res}² × {spp} | {tn[tree]} / {tn[treef]} | {t:.1f} / {tf:.1f} | {samples / t / 1e3:.0f} / {samples / tf / 1e3:.0f} | "
          f"{c['rays'] / t / 1e3:.0f} | {c['rays'] / c['samples']:.2f} | {c['inner_visits'] / c['traversals']:.1f} | {abytes / t / 1e6:.0f} | "
          f"{flops / t / 1e9:.1f} / {flopsf / tf / 1e9:.1f} | {c['redone_walks']:,} | {foot / 2**20:.0f} | {oc['samples'] / dt / 1e6:.2f} |", flush=True)


def main():
    print("| configuration | faces | launch | tree (exact / fast) | ms / launch (exact / fast) | Msamples/s (exact / fast) | Mrays/s (reference-equivalent, exact) | "
          "rays / sample | node visits / walk | algorithmic GB/s | useful fp32 TFLOP/s, ISA-counted (exact / fast) | walks repeated sequentially | "
          "device MiB | CPU oracle Msamples/s (%d threads, middle rows) |" % THREADS)
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    row("c1", 16, 5, 256)
    row("c2", 50, 5, 128)
    row("c3", 64, 5, 128)
    row("c4", 50, 5, 128)
    row("c5", 16, 3, 64)


if __name__ == "__main__":
    main()
