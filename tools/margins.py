#!/usr/bin/env python3
"""How far the box test and the triangle test disagree about one ray, measured on the CPU oracle over the committed
scenes: the evidence behind kCullSlack (2^-13), kNearHi (2^-12), kBand (1e-4) and kEdgeBand (2^-9)
of pathtracer_amd/csrc/core/kernel.h.   python3 tools/margins.py > profiles/r03_margins.md   (CPU only, minutes)"""
import math
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as ob  # noqa: E402
from pathtracer_amd import scenes  # noqa: E402
from tests import util  # noqa: E402


def lg(x):
    return "-" if x <= 0 else f"2^{math.log2(x):.1f}"


def main():
    threads = len(os.sched_getaffinity(0))
    with tempfile.TemporaryDirectory() as d:
        cases = [
            ("cornell.gltf 256x256x16", util.golden_scene("cornell"), 256, 16, (0, 0, 0)),
            ("dragon.gltf 384x384x8", util.golden_scene("dragon"), 384, 8, (0, 0, 0)),
            ("glass_mirror.gltf + env 384x384x8", util.golden_scene("glass_mirror", env=True), 384, 8, (1, 1, 1)),
            ("many_lights stand-in 256x256x8", util.load_bundle(scenes.write_many_lights_standin(d)), 256, 8, (0, 0, 0)),
            ("grid of 720,000 triangles 1e-3 long 512x512x4", util.load_bundle(scenes.write_synthetic_grid(d, name="mg", side=600)), 512, 4, (0, 0, 0)),
            ("grazing-incidence scene 200x152x8", util.grazing_scene(), 200, 8, (0, 0, 0)),
        ]
        print("| scene | rays | accepted hits seen while looking at leaves | box entry behind the triangle's t: count, largest gap | box exit in "
              "front of t: count, largest | leaf culled by distance though its triangle is nearer: count, largest gap | box MISSED though a "
              "triangle in it is hit: count, deepest such hit (smallest barycentric weight) | rays whose runner-up lies within 2^-12 / 2^-13 / "
              "2^-18 / 2^-23 of the winner |")
        print("|---|---|---|---|---|---|---|---|")
        worst = {"entry": 0.0, "exit": 0.0, "cull": 0.0, "rim": 0.0}
        for label, b, res, spp, clear in cases:
            h = res * 152 // 200 if "grazing" in label else res
            o = ob.Oracle(b.scene, b.bvh, b.lst, res, h, 0.0, tuple(float(c) for c in clear))
            m = o.margin_probe(42, spp, threads=threads)
            g = m["gap_below"]
            print(f"| {label} | {m['rays']:,} | {m['hits']:,} | {m['entry_behind']:,}, {lg(m['max_entry_behind'])} | {m['exit_in_front']:,}, "
                  f"{lg(m['max_exit_in_front'])} | {m['culled_by_distance']:,}, {lg(m['max_culled_gap'])} | {m['missed_box']:,} ({m['missed_box_inf_rinv']:,} by rays with a zero direction component), "
                  f"others: {lg(m['max_rim_of_missed_finite'])} at t <= {m['max_t_of_missed_finite']:.3g} | {g[4]:,} / {g[5]:,} / {g[10]:,} / {g[15]:,} of {m['rays_with_two_candidates']:,} |", flush=True)
            worst["entry"] = max(worst["entry"], m["max_entry_behind"])
            worst["exit"] = max(worst["exit"], m["max_exit_in_front"])
            worst["cull"] = max(worst["cull"], m["max_culled_gap"])
            worst["rim"] = max(worst["rim"], m["max_rim_of_missed_finite"])
        print()
        print(f"Largest over all scenes: box entry behind t by {lg(worst['entry'])} of t; box exit in front of t by {lg(worst['exit'])}; a leaf "
              f"culled by distance although its triangle was nearer by {lg(worst['cull'])} (kCullSlack = 2^-13, kNearHi = 2^-12); a hit whose leaf box "
              f"the ray missed lay at most {lg(worst['rim'])} inside its triangle (kEdgeBand = 2^-9).")


if __name__ == "__main__":
    main()
