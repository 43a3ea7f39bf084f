"""Manual soak (not collected by pytest): random scenes / sizes / seeds / sample counts / launch splits,
GPU image against the oracle, bit for bit. Usage: python3 tests/soak_gpu.py [n_cases] [seed] [only_case]
(only_case: draw the same sequence of cases but render just that one, on both trees, and say which pixels differ)
PT_SOAK_ARITHMETIC=fast renders the same cases with tolerance arithmetic instead and grades each by north_star's image
check: pixels non-finite on exactly one side, fraction of floats identical, per-pixel RMSE and mean relative error
against the oracle (a case fails when the mean differs by more than 2 % or a pixel is non-finite on one side only)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402
from pathtracer_amd import core, scenes  # noqa: E402
from tests import util  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    d = tempfile.mkdtemp()
    hdr = scenes.write_sky_hdr(d + "/sky.hdr")
    makers = [
        lambda k: (scenes.write_cornell_like(d, name=f"c{k}", detail=int(rng.randint(3, 14))), {}),
        lambda k: (scenes.write_many_lights_standin(d, name=f"m{k}", rings=int(rng.randint(8, 90)), segments=int(rng.randint(8, 120))), {}),
        lambda k: (scenes.write_dragon_standin(d, name=f"d{k}", rings=int(rng.randint(6, 70)), segments=int(rng.randint(6, 70))), {}),
        lambda k: (scenes.write_glass_mirror_standin(d, name=f"g{k}", detail=int(rng.randint(4, 30))), {"clear_texture": hdr}),
        lambda k: (scenes.write_textured_standin(d, name=f"t{k}", detail=int(rng.randint(4, 24))), {}),
    ]
    bad = 0
    worst = [0.0, 0.0, 1.0]  # fast mode: largest mean relative error, largest RMSE / mean, smallest identical fraction
    for k in range(n_cases):
        path, kw = makers[rng.randint(len(makers))](k)
        b = util.load_bundle(path, **kw)
        w, h = int(rng.randint(9, 200)), int(rng.randint(9, 160))
        spp, every = int(rng.randint(1, 24)), int(rng.randint(1, 12))
        seed, exposure = int(rng.randint(-5, 10 ** 6)), float(rng.choice([0.0, 0.5, -1.0, 2.0]))
        clear = tuple(float(x) for x in rng.choice([0.0, 0.3, 1.0], 3))
        os.environ["PT_TREE"] = str(rng.choice(["wide", "binary"]))
        os.environ["PT_TILE_ROWS"] = str(rng.choice(["8", "16"]))
        if only >= 0 and k != only:
            continue
        if only >= 0:
            cpu = ob.Oracle(b.scene, b.bvh, b.lst, w, h, exposure, clear).render(seed, spp, every)
            for tree in ("wide", "binary"):
                os.environ["PT_TREE"] = tree
                r = core.Renderer(0)
                r.upload(b.scene, b.bvh, b.lst)
                r.configure(w, h, exposure, clear)
                gpu = r.render(seed, spp, every)
                ne = (gpu != cpu) & ~(np.isnan(gpu) & np.isnan(cpu))
                ys, xs = np.nonzero(ne.any(axis=2))
                print(f"case {k} {os.path.basename(path)} {w}x{h} spp {spp}/{every} seed {seed} tree {tree}: "
                      f"{int(ne.sum())} floats differ", [(int(x), int(y), gpu[y, x].tolist(), cpu[y, x].tolist()) for y, x in zip(ys, xs)][:4])
                r.close()
            continue
        r = core.Renderer(0)
        fast = os.environ.get("PT_SOAK_ARITHMETIC") == "fast"
        if fast:
            r.set_arithmetic("fast")
        r.upload(b.scene, b.bvh, b.lst)
        r.configure(w, h, exposure, clear)
        gpu = r.render(seed, spp, every)
        cpu = ob.Oracle(b.scene, b.bvh, b.lst, w, h, exposure, clear).render(seed, spp, every)
        if fast:
            ok = np.isfinite(gpu) & np.isfinite(cpu)
            one_sided = int((np.isfinite(gpu) != np.isfinite(cpu)).sum())
            g, c = gpu[ok].astype(np.float64), cpu[ok].astype(np.float64)
            same = float((g == c).mean())
            rmse = float(np.sqrt(np.mean((g - c) ** 2)))
            mean = float(c.mean())
            rel_mean = abs(float(g.mean()) - mean) / max(mean, 1e-30)
            failed = one_sided != 0 or rel_mean > 0.02
            bad += failed
            worst[0], worst[1], worst[2] = max(worst[0], rel_mean), max(worst[1], rmse / max(mean, 1e-30)), min(worst[2], same)
            print(f"case {k}: {os.path.basename(path)} faces {len(b.scene.faces)} {w}x{h} spp {spp}/{every} seed {seed} tree {os.environ['PT_TREE']}: "
                  f"{same * 100:.1f} % identical, RMSE {rmse:.3g} ({rmse / max(mean, 1e-30):.3g} of the mean), mean relative error {rel_mean:.3g}, "
                  f"one-sided non-finite {one_sided}{' FAILED' if failed else ''}", flush=True)
            r.close()
            continue
        # a pixel that is NaN on both sides (degenerate shading frame, as in the reference) is the same pixel
        diff = int(((gpu != cpu) & ~(np.isnan(gpu) & np.isnan(cpu))).sum())
        bad += diff != 0
        print(f"case {k}: {os.path.basename(path)} faces {len(b.scene.faces)} {w}x{h} spp {spp}/{every} seed {seed} "
              f"tree {os.environ['PT_TREE']} rows {os.environ['PT_TILE_ROWS']}: {'OK' if diff == 0 else f'{diff} floats differ'}",
              flush=True)
        r.close()
    if os.environ.get("PT_SOAK_ARITHMETIC") == "fast":
        print(f"soak (tolerance arithmetic): {n_cases - bad} of {n_cases} cases within tolerance; largest mean relative error {worst[0]:.3g}, "
              f"largest RMSE / mean {worst[1]:.3g}, smallest identical fraction {worst[2]:.3f}")
        return 1 if bad else 0
    print("soak:", "all bit-identical" if bad == 0 else f"{bad} cases differ")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
