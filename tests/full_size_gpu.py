"""Manual check (not collected by pytest; about a minute): a BASELINE configuration at FULL size on the CPU
oracle and on the GPU with each form of the tree, compared bit for bit. The suite compares at sizes the
oracle finishes in seconds; this is the same comparison at the size the bench runs.

  python3 tests/full_size_gpu.py [--scene many_lights|dragon|glass|grid|dragon_standin|glass_standin] [--res 1024] [--spp 50] [--batches 4] [--threads 16]
  (dragon, glass = the reference's own dragon.gltf and glass_mirror.gltf + rose-garden environment from tests/golden/;
   grid = the synthetic 10 M-triangle scene of BASELINE config 5: --scene grid --res 4096 --spp 16 --batches 1.
   The gated suite runs one batch of each, tests/test_gpu_full_size.py; this script runs the configurations' full sample counts.)
"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402
from pathtracer_amd import core, scenes  # noqa: E402
from tests import util  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="many_lights")
    ap.add_argument("--res", type=int, default=1024)
    ap.add_argument("--spp", type=int, default=50)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--side", type=int, default=2236, help="grid scene: quads per side")
    a = ap.parse_args()
    clear = (1.0, 1.0, 1.0) if a.scene.startswith("glass") else (0.0, 0.0, 0.0)
    if a.scene == "dragon":
        b = util.golden_scene("dragon")
    elif a.scene == "glass":
        b = util.golden_scene("glass_mirror", env=True)
    else:
        gen = {"many_lights": scenes.write_many_lights_standin, "dragon_standin": scenes.write_dragon_standin,
               "glass_standin": scenes.write_glass_mirror_standin, "grid": lambda d: scenes.write_synthetic_grid(d, side=a.side)}[a.scene]
        with tempfile.TemporaryDirectory() as d:
            kw = {"clear_texture": scenes.write_sky_hdr(d + "/sky.hdr")} if a.scene == "glass_standin" else {}
            b = util.load_bundle(gen(d), **kw)
    total = a.spp * a.batches
    t0 = time.perf_counter()
    cpu = ob.Oracle(b.scene, b.bvh, b.lst, a.res, a.res, 0.0, clear).render(42, total, a.spp, threads=a.threads)
    dt = time.perf_counter() - t0
    print(f"oracle: {a.scene} {a.res}x{a.res} {total} spp in {dt:.1f} s on {a.threads} threads "
          f"({a.res * a.res * total / dt / 1e6:.2f} Msamples/s)", flush=True)
    bad = 0
    for tree in ("binary", "wide"):
        os.environ["PT_TREE"] = tree
        r = core.Renderer(0)
        r.upload(b.scene, b.bvh, b.lst)
        r.configure(a.res, a.res, 0.0, clear)
        gpu = r.render(42, total, a.spp)
        ms = r.last_kernel_ms()
        r.close()
        diff = (gpu.view(np.uint32) != cpu.view(np.uint32)) & ~(np.isnan(gpu) & np.isnan(cpu))  # a NaN on both sides is the same pixel
        n = int(diff.sum())
        both = np.isfinite(gpu) & np.isfinite(cpu)
        print(f"GPU {tree} tree ({ms:.1f} ms per launch): {n} of {diff.size} floats differ from the oracle in bits "
              f"({int(diff.any(axis=2).sum())} pixels, max abs {float(np.abs(np.where(both, gpu - cpu, 0)).max()):.3e}); "
              f"non-finite floats {int((~np.isfinite(gpu)).sum())} / {int((~np.isfinite(cpu)).sum())}", flush=True)
        bad += n
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
