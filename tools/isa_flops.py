#!/usr/bin/env python3
"""fp32 operations per lane of each code region of the render kernel, counted once from the ISA.

  python3 tools/isa_flops.py exact|fast profiles/rNN_isa_flops_<arithmetic>.json

Compiles core/ptcore.hip (exact) or core/kernels_fast.hip (fast) for gfx950 with -DPT_MARK_REGIONS -S (device code
only): kernel.h then leaves comment marks in the assembly around the four-wide node step, the binary node step, the
test of a leaf's pair of triangles, the first ray of a sample, the shadow-result step and the path-vertex handler
(brdf_eval_pair is out of line and counted as its own function). Every instruction between a region's marks is
classified and weighted: fma / mac = 2, packed = two lanes' worth, add / sub / mul / min / max / compare / convert /
reciprocal / root / log / sin / cos and the division helpers = 1, min3 / max3 = 2; moves, selects, integer and
address arithmetic = 0. Limits, stated: marks do not stop the scheduler from moving an instruction across them, and a
region with branches (the vertex handler: miss / hit, three kinds of light, two rejection loops) is counted along ALL
its paths once, so its figure is an upper bound on one call; the node steps and the leaf test are straight-line code.
bench.py multiplies these by the per-lane executions of each region that the instrumented launch counts.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pathtracer_amd", "csrc")

W2 = re.compile(r"v_(fma|fmac|mac|mad|fmaak|fmamk|madak|madmk)_f32")
W_PK2 = re.compile(r"v_pk_(add|mul|min|max)_f32")
W_PK4 = re.compile(r"v_pk_fma_f32")
W1 = re.compile(r"v_(add|sub|subrev|mul|min|max|rcp|rsq|sqrt|log|exp|sin|cos|floor|fract|trunc|rndne|ceil|ldexp|frexp_mant|div_scale|div_fixup|mul_legacy)_f32"
                r"|v_cmpx?_[a-z]+_f32|v_cvt_f32_[ui]32|v_cvt_[ui]32_f32|v_cvt_f32_ubyte\d")
W1_3 = re.compile(r"v_(min3|max3|med3)_f32")
W_DIVFMAS = re.compile(r"v_div_fmas_f32")
F64 = re.compile(r"v_[a-z0-9_]+_f64")


def classify(line):
    m = re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", line.split()[0])
    if W_PK4.fullmatch(m):
        return 4
    if W_PK2.fullmatch(m):
        return 2
    if W2.fullmatch(m) or W1_3.fullmatch(m) or W_DIVFMAS.fullmatch(m):
        return 2
    if W1.fullmatch(m):
        return 1
    return 0


def function_body(asm, name):
    i = asm.index("\n" + name + ":")
    j = asm.index(".Lfunc_end", i)
    return asm[i:j]


def count(lines):
    flop = valu = total = f64 = 0
    for ln in lines:
        t = ln.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        total += 1
        if t.startswith("v_"):
            valu += 1
            flop += classify(t)
            if F64.match(t.split()[0]):
                f64 += 1
    return {"fp32_ops": flop, "valu_instructions": valu, "instructions": total, "f64_instructions": f64}


def regions(body):
    out, open_at = {}, {}
    lines = body.split("\n")
    for k, ln in enumerate(lines):
        m = re.search(r"; PT_MARK (\w+) (begin|end)", ln)
        if not m:
            continue
        name, what = m.groups()
        if what == "begin":
            open_at[name] = k
        elif name in open_at:
            c = count(lines[open_at.pop(name):k])
            if name not in out or c["fp32_ops"] > out[name]["fp32_ops"]:  # (a region the compiler duplicated: the larger copy)
                out[name] = c
    return out


def main():
    arithmetic, out_path = sys.argv[1], sys.argv[2]
    src = "core/kernels_fast.hip" if arithmetic == "fast" else "core/ptcore.hip"
    flags = ["-DPT_FAST=1", "-ffp-contract=fast"] if arithmetic == "fast" else ["-ffp-contract=off"]
    ns = "fast" if arithmetic == "fast" else "exact"
    with tempfile.TemporaryDirectory() as d:
        s_path = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-fno-fast-math", "-DPT_MARK_REGIONS",
                               "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", s_path, src] + flags, cwd=CSRC,
                              stderr=subprocess.DEVNULL)
        asm = open(s_path).read()
    n = len(ns)
    kernels = {"wide": f"_ZN3ptc{n}{ns}13render_kernelILb0ELi48ELi8ELb1EEEvNS_12KernelParamsEPfPy",
               "binary": f"_ZN3ptc{n}{ns}13render_kernelILb0ELi32ELi8ELb0EEEvNS_12KernelParamsEPfPy"}
    res = {"arithmetic": arithmetic, "command": "python3 tools/isa_flops.py " + arithmetic, "weights": __doc__.split("classified and weighted: ")[1].split(". Limits")[0],
           "kernels": {}}
    for key, name in kernels.items():
        body = function_body(asm, name)
        res["kernels"][key] = {"symbol": name, "whole_kernel": count(body.split("\n")), "regions": regions(body)}
    pair = [m for m in re.findall(r"\n(_ZN3ptc\d+\w+?14brdf_eval_pair\w+):", asm)]
    res["brdf_eval_pair"] = count(function_body(asm, pair[0]).split("\n")) if pair else None
    w, b = res["kernels"]["wide"]["regions"], res["kernels"]["binary"]["regions"]
    pair_ops = res["brdf_eval_pair"]["fp32_ops"] if res["brdf_eval_pair"] else 0
    res["flop_per_lane"] = {"wide_step": w["wide_step"]["fp32_ops"], "binary_step": b["binary_step"]["fp32_ops"],
                            "leaf_pair": w["leaf_pair"]["fp32_ops"], "start_sample": w["start_sample"]["fp32_ops"],
                            "shadow_result": w.get("shadow_result", {"fp32_ops": 0})["fp32_ops"],
                            "vertex_handler": w["vertex_handler"]["fp32_ops"], "brdf_eval_pair": pair_ops}
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res["flop_per_lane"]), {k: v["regions"].keys() for k, v in res["kernels"].items()})
    for k, v in w.items():
        print(" wide kernel", k, v)
    print(" binary kernel binary_step", b["binary_step"], "brdf_eval_pair", res["brdf_eval_pair"])


if __name__ == "__main__":
    main()
