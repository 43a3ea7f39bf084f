#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the radiance kernel on BASELINE.json's configurations.

  python bench.py --gpus N --steps K --warmup W [--config c2|c3|c5|c1|c4] [--weak]

The JOB is fixed (strong scaling, the reference's own metric: README.md:29-59, batch loop src/main.cu:101-133):
K batches ("steps") of the configuration's samples-every-update, batch b seeded 42 + b, rendered in index order on
one GPU, or sharded b mod N over N GPUs (one rank per GPU, launched by torch.distributed.run) with the per-rank
accumulators combined by ONE RCCL sum all-reduce inside the timed region. value = W*H*spp_total / elapsed, the same
job at every N. Batches that do not fill a whole round of N are cut into N equal parts, so every rank renders the same
number of samples (c2 at N = 8: 8 batches of 25 spp instead of 4 of 50, as SURVEY.md section 8e prescribes; the line
lists the sizes in config.batch_sizes). --weak keeps round 2's behaviour: K batches PER RANK.

  c2 (default)  many_lights STAND-IN 1024x1024, batches of 50 spp (K = 4: the README's 200 spp)
  c3            the reference's dragon.gltf (golden arrays) 1024x1024, batches of 64 spp (K = 8: 512 spp)
  c5            synthetic 10 M-triangle grid 4096x4096, batches of 128 spp (K = 8: 1024 spp)
  c1, c4        cornell 256x256x16 (plumbing) and glass_mirror + env 1024x1024, batches of 32 spp

Inputs are resident in HBM before the timed region. Rank 0 prints one JSON line with the throughput, a roofline
object for the dominant kernel and (N = 1) a cpu_baseline object: the CPU oracle timed on a bounded sample of the
same workload.
"""
import argparse
import glob
import json
import os
import re
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 42
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
VALU_PEAK_TFLOPS = 157.3   # fp32 vector peak: 256 CUs x 4 SIMDs x 16 lanes x 2 (packed) x 2 (fma) x 2.4 GHz
VALU_PEAK_NOPK_NOFMA = VALU_PEAK_TFLOPS / 4.0  # one scalar-form add or mul per lane per cycle: what exact arithmetic may use at best

CONFIGS = {
    "c1": {"res": 256, "spp": 16, "steps": 1, "clear": (0.0, 0.0, 0.0), "cpu_rows": 256, "profile_key": "cornell",
           "label": "cornell.gltf (the reference's asset, golden arrays; 1,940 tris)"},
    "c2": {"res": 1024, "spp": 50, "steps": 4, "clear": (0.0, 0.0, 0.0), "cpu_rows": 256, "profile_key": "many_lights",
           "label": "many_lights STAND-IN (room+point+sun+emitter, 322804 tris)"},
    "c3": {"res": 1024, "spp": 64, "steps": 8, "clear": (0.0, 0.0, 0.0), "cpu_rows": 128, "profile_key": "dragon",
           "label": "dragon.gltf (the reference's asset, golden arrays; 139,482 tris)"},
    "c4": {"res": 1024, "spp": 32, "steps": 8, "clear": (1.0, 1.0, 1.0), "cpu_rows": 256, "profile_key": "glass",
           "label": "glass_mirror.gltf (golden arrays) + sunny_rose_garden env point-sampled 4x down to 256x128"},
    "c5": {"res": 4096, "spp": 128, "steps": 8, "clear": (0.0, 0.0, 0.0), "cpu_rows": 8, "profile_key": "grid",
           "label": "synthetic 10 M-triangle grid (2236^2 displaced quads + 8 emitters)"},
}


def load_config_scene(name):
    """(scene, bvh, lst) of a configuration through the host pipeline (not timed)."""
    from pathtracer_amd import scenes
    from tests import util
    with tempfile.TemporaryDirectory() as d:
        if name == "c1":
            b = util.golden_scene("cornell")
        elif name == "c2":
            b = util.load_bundle(scenes.write_many_lights_standin(d))
        elif name == "c3":
            b = util.golden_scene("dragon")
        elif name == "c4":
            b = util.golden_scene("glass_mirror", env=True)
        else:
            b = util.load_bundle(scenes.write_synthetic_grid(d, side=2236))
    return b.scene, b.bvh, b.lst


def plan_job(steps, spp_step, world, weak):
    """Samples per batch of every batch of the whole job, in batch order (batch b runs on rank b mod world with seed
    42 + b). Strong: the job is steps x spp_step samples per pixel whatever the number of ranks. Whole rounds of `world`
    batches keep the configuration's samples-every-update; the batches left over (steps mod world of them, all of them
    when steps < world) are cut into `world` equal parts, one per rank, so that every rank renders the same number of
    samples (c2 at N = 8: 8 batches of 25 spp instead of 4 of 50; 20 steps at N = 8: 16 batches of
    50 + 8 of 25). A remainder that does not divide leaves parts that differ by one sample. Weak: steps batches per rank."""
    if weak:
        return [spp_step] * (steps * world)
    sizes = [spp_step] * (steps // world * world)
    left = (steps % world) * spp_step
    if left:
        sizes += [s for s in (left // world + (1 if r < left % world else 0) for r in range(world)) if s > 0]
    return sizes


def plan_rank(sizes, rank, world):
    """[(batch index, seed offset, samples)] of this rank, in launch order: render.plan_batches for batches of given sizes."""
    return [(b, b, n) for b, n in enumerate(sizes) if b % world == rank]


def algorithmic_bytes(c, width, height, tree=2):
    """64 B per inner-node visit of the binary tree (two child boxes + two references); a visit
    of the four-wide form loads 112 B (four boxes = 96 B + four references = 16 B: the 7 dwordx4 loads of
    walk_slice; the record is padded to 128 B, the padding is not fetched). 40 B per triangle test (positions
    + id), 240 B per shaded hit (face + 3 x (normal, tangent, uv) + material), 24 B per pixel per launch
    (accumulator read + write)."""
    node = 112 if tree == 4 else 64
    return node * c["inner_visits"] + 40 * c["tri_tests"] + 240 * c["shaded_hits"] + 24 * width * height


def algorithmic_flops(c, tree, table):
    """fp32 operations the useful lanes of one launch execute: per-lane executions of each code region (instrumented
    launch) x the region's fp32 operations per lane counted once from the ISA (profiles/*_isa_flops.json, made by
    tools/isa_flops.py: v_fma = 2, packed forms = 2 lanes' worth, compares / min / max / conversions = 1)."""
    if not table:
        return None, None
    per = table["flop_per_lane"]
    node = per["wide_step"] if tree == 4 else per["binary_step"]
    parts = {"node_steps": node * c["inner_visits"], "leaf_tests": per["leaf_pair"] * c["leaf_tests"],
             "vertex_handler": per["vertex_handler"] * c["handler_calls"] + per["brdf_eval_pair"] * c["bounces"],
             "camera_rays": per["start_sample"] * c["samples"], "shadow_results": per["shadow_result"] * c["shadow_results"]}
    return sum(parts.values()), parts


def cpu_baseline(cfg, scene, bvh, lst, threads):
    """The oracle ("port") on the middle image rows of one batch of the same workload."""
    import numpy as np
    from oracle import binding as ob
    res, spp = cfg["res"], cfg["spp"]
    o = ob.Oracle(scene, bvh, lst, res, res, 0.0, cfg["clear"])
    half = min(cfg["cpu_rows"], res) // 2
    rows = (res // 2 - half, res // 2 + half)
    img = np.zeros((res, res, 3), np.float32)
    t0 = time.perf_counter()
    _, ctr = o.render_batch(SEED, 0, spp, img, threads=threads, rows=rows)
    dt = time.perf_counter() - t0
    return {"value": round(ctr["samples"] / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"rows {rows[0]}..{rows[1] - 1} of one {res}x{res}x{spp}spp batch of the same workload "
                      f"({ctr['samples']} samples, {ctr['rays']} rays, {dt:.2f} s wall)",
            "mrays_per_s": round(ctr["rays"] / dt / 1e6, 3)}


def newest_profile(pattern, key=None):
    """The newest committed profile summary matching `pattern` (round, then kernel version), optionally of one
    workload (its name appears in the file name)."""
    files = [f for f in glob.glob(os.path.join(ROOT, "profiles", pattern)) if key is None or key in os.path.basename(f)]
    files.sort(key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    for f in reversed(files):
        try:
            return json.load(open(f)), os.path.join("profiles", os.path.basename(f))
        except (OSError, ValueError):
            continue
    return {}, None


def pmc_profile(key, arithmetic):
    """Hardware counters cannot be read from inside this process, so what the bench line quotes from here (fabric
    traffic, VALU busy, lane utilisation) is the PROFILED launch of the same workload (separate rocprofv3 --pmc
    passes, tools/profile_round.sh + tools/summarize_profiles.py), named by its file, not a live figure."""
    tag = "fast" if arithmetic == "fast" else "exact"
    d, src = newest_profile(f"r*_pmc_*{tag}*.json", key)
    if not d and tag == "exact":
        d, src = newest_profile("r*_pmc_*.json", key)
        if src and "fast" in src:
            d, src = {}, None
    return d.get("derived", {}), src


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="batches in the whole job (per rank with --weak); default: the configuration's")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--weak", action="store_true", help="K batches per rank instead of K in total")
    ap.add_argument("--arithmetic", default="exact", choices=("exact", "fast"),
                    help="exact: bit-identical to the CPU oracle (the gate); fast: tolerance arithmetic, measured beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tolerance", action="store_true", help="skip the untimed tolerance-arithmetic launches")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"), help="gloo: rehearsal of N ranks on one GPU")
    ap.add_argument("--one-gpu", action="store_true", help="every rank uses device 0 (with --backend gloo)")
    ap.add_argument("--res", type=int, default=None, help="override the image size (tests)")
    ap.add_argument("--spp", type=int, default=None, help="override the samples per batch (tests)")
    ap.add_argument("--dump-image", default=None, help="rank 0 writes the combined image here as .npy (tests)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from pathtracer_amd import core, render

    cfg = dict(CONFIGS[args.config])
    if args.res:
        cfg["res"] = args.res
    if args.spp:
        cfg["spp"] = args.spp
    steps = args.steps if args.steps is not None else cfg["steps"]
    if steps <= 0:
        raise SystemExit("--steps must be positive")
    width = height = cfg["res"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N with N > 1 must be launched by torch.distributed.run (one rank per GPU)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # launched by torch.distributed.run (it exports WORLD_SIZE): one rank per GPU over RCCL, also when N = 1
    use_dist = world > 1 or "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    # ---- inputs (host pipeline; not timed) ----------------------------------------------
    scene, bvh, lst = load_config_scene(args.config)
    r = core.Renderer(local_rank)
    r.set_arithmetic(args.arithmetic)
    r.upload(scene, bvh, lst)
    r.configure(width, height, 0.0, cfg["clear"])
    # everything that touches the accumulator runs on ONE side stream: the render launches (handed to the
    # library by handle), the weighting and the RCCL reduce (torch orders it against the current stream)
    side = torch.cuda.Stream(device)
    torch.cuda.set_stream(side)
    acc = torch.zeros(width * height * 3, dtype=torch.float32, device=device)
    stream = side.cuda_stream
    assert stream != 0

    def barrier():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    # ---- the job: which batches exist, and which are this rank's ------------------------------
    sizes = plan_job(steps, cfg["spp"], world, args.weak)
    spp_total, n_batches, spp_batch = sum(sizes), len(sizes), sizes[0]
    plan = plan_rank(sizes, rank, world)  # [(b, seed offset, samples)]: batch b on rank b mod N

    # ---- warmup (untimed; the first launch also times both forms of the tree and records tile costs) ----
    for i in range(args.warmup):
        r.render_batch_into(acc.data_ptr(), SEED + 1000 + i, 0, spp_batch, stream)
    if use_dist:
        dist.all_reduce(acc)
    acc.zero_()
    barrier()

    # ---- timed region: this rank's batches + one reduce -----------------------------------------
    t0 = time.perf_counter()
    local_spp = render.render_local(r, acc, SEED, plan, stream)
    render.combine(acc, local_spp, spp_total)
    barrier()
    elapsed = time.perf_counter() - t0
    # HIP events the library recorded around every launch, on the stream it was launched on
    # (launch k back is plan[-1 - k]; the per-batch figures below are of the batches of batch 0's size)
    kernel_ms = [r.kernel_ms(k) for k in range(min(len(plan), 64)) if plan[-1 - k][2] == spp_batch]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    finite = bool(torch.isfinite(acc).all().item())
    mean = float(acc[torch.isfinite(acc)].mean().item())
    if args.dump_image and rank == 0:
        import numpy as np
        np.save(args.dump_image, acc.view(height, width, 3).cpu().numpy())

    if rank == 0:
        samples = width * height * spp_total
        counters = r.count_batch(SEED, spp_batch)  # instrumented variant, first timed batch, untimed
        avg_ms = sum(kernel_ms) / len(kernel_ms) if kernel_ms else float("nan")
        tree = r.walk_tree()
        abytes = algorithmic_bytes(counters, width, height, tree)
        achieved = abytes / (avg_ms * 1e-3) / 1e9
        prof, prof_src = pmc_profile(cfg["profile_key"], args.arithmetic)
        # the profiled launch may have fewer samples per pixel than this batch (c5 is profiled at 16 spp, one GPU's share
        # of a 128-spp batch): per-launch counter totals scale with the samples, ratios do not
        m = re.search(r"_(\d+)spp", prof_src or "")
        prof_scale = spp_batch / float(m.group(1)) if m else 1.0
        traffic = int(prof["hbm_traffic_per_launch"] * prof_scale) if "hbm_traffic_per_launch" in prof else None
        isa, isa_src = newest_profile(f"r*_isa_flops_{args.arithmetic}*.json")
        flops, flop_parts = algorithmic_flops(counters, tree, isa)
        tflops = flops / (avg_ms * 1e-3) / 1e12 if flops else None
        # lane utilisation of the instrumented launch, weighted by the time the waves spent in each region
        lu, ts = counters.get("lane_utilisation", {}), counters.get("time_share", {})
        live_lanes = None
        if ts and lu:
            parts = [("node_phases", "inner"), ("leaf_phases", "leaf"), ("vertex_handler", "main"), ("shadow_results", "nee")]
            w = sum(ts[a] for a, _ in parts)
            live_lanes = round(sum(ts[a] * (lu[b] or 0.0) for a, b in parts) / w, 4) if w else None
        kname = "render_kernel<COUNT=false, STACK, ROWS=%d, WIDE=%s> (%s arithmetic)" % (
            16 if width * height >= 3_000_000 else 8, "true" if tree == 4 else "false", args.arithmetic)
        parity = ("bit-identical to the CPU oracle (tests/test_gpu_full_size.py); oracle vs cuRAND: XORWOW scramble constants and "
                  "Box-Muller float path unverified (cuRAND absent from the image)") if args.arithmetic == "exact" else (
                  "tolerance arithmetic: u32 RNG stream bit-exact, primary-hit ids and image error against the oracle stated in "
                  "tests/test_gpu_tolerance.py and DESIGN.md section 3")
        out = {
            "metric": f"Msamples/s (node) {cfg['label'].split(' (')[0]} {width}x{height}, {spp_total} spp in {n_batches} batches",
            "value": round(samples / elapsed / 1e6, 3),
            "unit": "Msamples/s", "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak" if args.weak else "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['label']} {width}x{height}, {spp_total} spp in {n_batches} batches of "
                                   f"{'/'.join(str(x) for x in sorted(set(sizes), reverse=True))} spp, seed 42+batch, batch b on rank b mod {world}",
                       "resolution": [width, height], "spp_per_batch": spp_batch, "batch_sizes": sizes if len(sizes) <= 64 else sizes[:64] + ["..."],
                       "batches": n_batches, "spp_total": spp_total, "spp_on_rank0": sum(n for _, _, n in plan),
                       "batches_on_rank0": len(plan), "parallelism": f"sample-shard x{world}", "arithmetic": args.arithmetic,
                       "walk_tree": tree, "bvh_nodes": int(bvh.node_count), "faces": int(len(scene.faces)), "lst_nodes": int(len(lst.nodes)),
                       "parity": parity},
            "elapsed_s": round(elapsed, 4),
            "kernel_ms_per_batch": round(avg_ms, 3),
            "kernel_ms_per_step": round(avg_ms * cfg["spp"] / spp_batch, 3),
            "kernel_msamples_per_s": round(width * height * spp_batch / (avg_ms * 1e-3) / 1e6, 3),
            "mrays_per_s": round(counters["rays"] / (avg_ms * 1e-3) / 1e6 * world, 2),
            "mtraversals_per_s": round(counters["traversals"] / (avg_ms * 1e-3) / 1e6 * world, 2),
            "counters_per_batch": counters,
            "image_finite": finite, "image_mean": round(mean, 6),
            # What binds: VALU issue under divergence. achieved = fp32 operations of the useful
            # lanes (per-lane region executions of the instrumented launch x fp32 operations per lane from the ISA) /
            # the launch's HIP-event time, against the packed-FMA vector peak. busy / lane_utilisation: hardware
            # counters of the profiled launch of the same workload (wave-level VALU instructions x 4 cycles / (1024
            # SIMDs x launch cycles); SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)).
            "roofline": {"bound": "valu", "achieved": round(tflops, 3) if tflops else None, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tflops / VALU_PEAK_TFLOPS, 5) if tflops else None,
                         "frac_of_scalar_form_peak": round(tflops / VALU_PEAK_NOPK_NOFMA, 5) if tflops else None,
                         "scalar_form_peak": VALU_PEAK_NOPK_NOFMA,
                         "traffic": traffic, "kernel": kname, "flops_per_launch": int(flops) if flops else None,
                         "flops_by_region": {k: int(v) for k, v in flop_parts.items()} if flop_parts else None,
                         "flop_per_lane_source": isa_src,
                         "busy": round(prof["valu_busy"], 4) if "valu_busy" in prof else None,
                         "lane_utilisation": round(prof["valu_lane_utilisation"], 4) if "valu_lane_utilisation" in prof else None,
                         "useful_issue_frac": round(prof["valu_useful_issue"], 4) if "valu_useful_issue" in prof else None,
                         "wave_instructions_per_launch": int(prof["valu_wave_instructions"] * prof_scale) if "valu_wave_instructions" in prof else None,
                         "counters_scaled_by": round(prof_scale, 4),
                         "l2_hit_rate": round(prof["l2_hit_rate"], 4) if "l2_hit_rate" in prof else None,
                         # the same quantity from gfx950's own FLOPS counters (SQ_INSTS_VALU_FLOPS_FP32 + _TRANS of the profiled
                         # launch: add / mul / fma / transcendental only, no compares or min / max): what the SIMDs issued for
                         # all 64 lanes, and the share the active lanes computed
                         "hw_counter_tflops_issued": round(prof["fp32_tflops_issued"], 3) if "fp32_tflops_issued" in prof else None,
                         "hw_counter_tflops_active_lanes": round(prof["fp32_tflops_active_lanes"], 3) if "fp32_tflops_active_lanes" in prof else None,
                         "resident_waves_per_simd": round(prof["resident_waves_per_simd"], 3) if "resident_waves_per_simd" in prof else None,
                         "wave_cycle_shares": {k: round(v, 4) for k, v in prof["wave_cycle_shares"].items()} if "wave_cycle_shares" in prof else None,
                         "counters_source": prof_src, "live_lane_utilisation_instrumented": live_lanes},
            # Contract form: algorithmic bytes / launch time against the HBM peak. The bytes are served by
            # L2 / Infinity Cache (fabric traffic is a fraction of them), so this fraction does not bind.
            "roofline_hbm": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": prof_src,
                             "traffic_over_algorithmic": round(traffic / abytes, 4) if traffic else None,
                             "kernel": kname, "algorithmic_bytes_per_launch": int(abytes),
                             "bytes_per_unit": {"inner_visit": 112 if tree == 4 else 64, "triangle_test": 40, "shaded_hit": 240, "pixel": 24}},
        }
        if not args.no_tolerance and args.arithmetic == "exact":
            # The same batches with tolerance arithmetic (contraction on, hardware reciprocal / root, no tie
            # bookkeeping), untimed by the wall clock: HIP-event time per launch beside the exact kernel's.
            r.set_arithmetic("fast")
            scratch = torch.zeros_like(acc)
            n_tol = min(len(plan), 4)
            r.render_batch_into(scratch.data_ptr(), SEED + 2000, 0, spp_batch, stream)  # tree timing + tile costs
            for k in range(n_tol):
                r.render_batch_into(scratch.data_ptr(), SEED + plan[k][1], 0, spp_batch, stream)
            torch.cuda.synchronize(device)
            tol_ms = sum(r.kernel_ms(k) for k in range(n_tol)) / n_tol
            out["kernel_ms_per_batch_tolerance"] = round(tol_ms, 3)
            out["kernel_ms_per_step_tolerance"] = round(tol_ms * cfg["spp"] / spp_batch, 3)
            out["kernel_msamples_per_s_tolerance"] = round(width * height * spp_batch / (tol_ms * 1e-3) / 1e6, 3)
            out["tolerance"] = {"arithmetic": "fp contraction on (fused slab b*rinv - o*rinv, dots, crosses, Moeller-Trumbore), v_rcp_f32 / "
                                              "v_sqrt_f32 / v_rsq_f32 (1 ulp) for division and roots, v_log/v_sin/v_cos in Box-Muller, "
                                              "no second-candidate bookkeeping or sequential re-walks; RNG integer stream untouched",
                                "walk_tree": r.walk_tree(), "stated_in": "tests/test_gpu_tolerance.py, DESIGN.md section 3"}
            r.set_arithmetic("exact")
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            # every core this process may use: its affinity mask, and the share the GPU box assigns to one GPU's job
            # (it exports OMP_NUM_THREADS = 16 and kills jobs that oversubscribe; os.cpu_count() there is the whole
            # 8-GPU machine)
            allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            threads = max(1, min(allowed, int(os.environ.get("OMP_NUM_THREADS", str(allowed)))))
            out["cpu_baseline"] = cpu_baseline(cfg, scene, bvh, lst, threads)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        print(json.dumps(out), flush=True)

    r.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
