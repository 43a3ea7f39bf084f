#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the radiance kernel on the many_lights configuration.

  python bench.py --gpus N --steps K --warmup W

A "step" is one launch of the hot path over one batch: 1024x1024 pixels x 50 spp of the
many_lights STAND-IN scene (the reference's many_lights.bin does not ship; the stand-in has the
same room, lights, emitter, camera, materials and a 322 792-triangle mesh in the scan's bounds —
pathtracer_amd/scenes.py). K=4 is the README run (200 spp in batches of 50, seed 42). Inputs are
resident in HBM before the timed region. With N > 1 (one rank per GPU, launched by
torch.distributed.run) every rank renders K batches of its own (weak scaling), batch b = j*N + rank
seeded 42 + b, and the per-rank accumulators are combined by ONE RCCL sum all-reduce inside the
timed region.

Rank 0 prints one JSON line with the throughput, a roofline object for the dominant kernel
(algorithmic bytes from the instrumented kernel variant / HIP-event launch time) and a
cpu_baseline object (the CPU oracle timed on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WIDTH = HEIGHT = 1024
SPP_PER_STEP = 50
SEED = 42
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(c, width, height, tree=2):
    """64 B per inner-node visit of the binary tree (two child boxes + two references); a visit
    of the four-wide form loads 112 B (four boxes = 96 B + four references = 16 B: the 7 dwordx4 loads of
    walk_slice; the record is padded to 128 B, the padding is not fetched). 40 B per triangle test (positions
    + id), 240 B per shaded hit (face + 3 x (normal, tangent, uv) + material), 24 B per pixel per launch
    (accumulator read + write)."""
    node = 112 if tree == 4 else 64
    return node * c["inner_visits"] + 40 * c["tri_tests"] + 240 * c["shaded_hits"] + 24 * width * height


def cpu_baseline(scene, bvh, lst, threads):
    """The oracle ("port") on the 256 middle image rows x 1024 px x 50 spp of the same workload (a quarter of a step)."""
    from oracle import binding as ob
    import numpy as np
    o = ob.Oracle(scene, bvh, lst, WIDTH, HEIGHT)
    rows = (HEIGHT // 2 - 128, HEIGHT // 2 + 128)
    img = np.zeros((HEIGHT, WIDTH, 3), np.float32)
    t0 = time.perf_counter()
    _, ctr = o.render_batch(SEED, 0, SPP_PER_STEP, img, threads=threads, rows=rows)
    dt = time.perf_counter() - t0
    return {"value": round(ctr["samples"] / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
            "sample": f"rows {rows[0]}..{rows[1] - 1} of the same 1024x1024x50spp batch "
                      f"({ctr['samples']} samples, {ctr['rays']} rays, {dt:.2f} s wall)",
            "mrays_per_s": round(ctr["rays"] / dt / 1e6, 3)}


def pmc_profile():
    """The newest committed PMC summary of this workload's render kernel (profiles/rNN_pmc_*.json: separate
    rocprofv3 --pmc passes, made by tools/profile_round.sh + tools/summarize_profiles.py). Hardware counters
    cannot be read from inside this process, so what the bench line quotes from here (fabric traffic, VALU
    busy, lane utilisation) is the PROFILED launch of the same command, named by its file, not a live figure."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_*.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])  # round, then kernel version
    for f in reversed(files):
        try:
            d = json.load(open(f)).get("derived", {})
            if "hbm_traffic_per_launch" in d:
                return d, os.path.join("profiles", os.path.basename(f))
        except (OSError, ValueError):
            continue
    return {}, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from pathtracer_amd import core, host, render, scenes

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    # ---- inputs (host pipeline; not timed) ----------------------------------------------
    with tempfile.TemporaryDirectory() as d:
        scene = host.Scene.load_gltf(scenes.write_many_lights_standin(d))
    bvh, lst = host.BVH(scene), host.LST(scene)
    r = core.Renderer(local_rank)
    r.upload(scene, bvh, lst)
    r.configure(WIDTH, HEIGHT, 0.0, (0.0, 0.0, 0.0))
    # everything that touches the accumulator runs on ONE side stream: the render launches (handed to the
    # library by handle), the 1/N scaling and the RCCL reduce (torch orders it against the current stream)
    side = torch.cuda.Stream(device)
    torch.cuda.set_stream(side)
    acc = torch.zeros(WIDTH * HEIGHT * 3, dtype=torch.float32, device=device)
    stream = side.cuda_stream
    assert stream != 0

    def barrier():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    # ---- warmup ----------------------------------------------------------------------------
    for i in range(args.warmup):
        r.render_batch_into(acc.data_ptr(), SEED + 1000 + i, 0, SPP_PER_STEP, stream)
    if use_dist:
        dist.all_reduce(acc)
    acc.zero_()
    barrier()

    # ---- timed region: K batches per rank + one reduce ---------------------------------------
    t0 = time.perf_counter()
    prev = 0
    for j in range(args.steps):
        b = j * world + rank
        r.render_batch_into(acc.data_ptr(), SEED + b, prev, SPP_PER_STEP, stream)
        prev += SPP_PER_STEP
    render.combine(acc, prev, prev * world)
    barrier()
    elapsed = time.perf_counter() - t0
    # HIP events the library recorded around every launch, on the stream it was launched on
    kernel_ms = [r.kernel_ms(k) for k in range(min(args.steps, 64))]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    finite = bool(torch.isfinite(acc).all().item())
    mean = float(acc.mean().item())

    if rank == 0:
        spp_total = SPP_PER_STEP * args.steps * world
        samples = WIDTH * HEIGHT * spp_total
        counters = r.count_batch(SEED, SPP_PER_STEP)  # instrumented variant, first timed batch, untimed
        avg_ms = sum(kernel_ms) / len(kernel_ms)
        tree = r.walk_tree()
        abytes = algorithmic_bytes(counters, WIDTH, HEIGHT, tree)
        achieved = abytes / (avg_ms * 1e-3) / 1e9
        prof, prof_src = pmc_profile()
        traffic = int(prof["hbm_traffic_per_launch"]) if "hbm_traffic_per_launch" in prof else None
        # lane utilisation of the instrumented launch, weighted by the time the waves spent in each region
        lu, ts = counters.get("lane_utilisation", {}), counters.get("time_share", {})
        live_lanes = None
        if ts and lu:
            parts = [("node_phases", "inner"), ("leaf_phases", "leaf"), ("vertex_handler", "main"), ("shadow_results", "nee")]
            w = sum(ts[a] for a, _ in parts)
            live_lanes = round(sum(ts[a] * (lu[b] or 0.0) for a, b in parts) / w, 4) if w else None
        out = {
            "metric": f"Msamples/s (node) many_lights STAND-IN 1024x1024, {spp_total} spp in steps of {SPP_PER_STEP}",
            "value": round(samples / elapsed / 1e6, 3),
            "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "many_lights STAND-IN (room+point+sun+emitter, 322804 tris) 1024x1024, "
                                   f"{SPP_PER_STEP} spp per step, seed 42+batch", "resolution": [WIDTH, HEIGHT],
                       "spp_per_step": SPP_PER_STEP, "spp_total": spp_total, "parallelism": f"sample-shard x{world}",
                       "walk_tree": tree, "bvh_nodes": int(bvh.node_count), "faces": int(len(scene.faces)), "lst_nodes": int(len(lst.nodes)),
                       "parity": "bit-identical to the CPU oracle (tests/test_gpu_full_size.py); oracle vs cuRAND: XORWOW "
                                 "scramble constants and Box-Muller float path unverified (cuRAND absent from the image)"},
            "kernel_ms_per_step": round(avg_ms, 3),
            "kernel_msamples_per_s": round(WIDTH * HEIGHT * SPP_PER_STEP / (avg_ms * 1e-3) / 1e6, 3),
            "mrays_per_s": round(counters["rays"] / (avg_ms * 1e-3) / 1e6 * world, 2),
            "mtraversals_per_s": round(counters["traversals"] / (avg_ms * 1e-3) / 1e6 * world, 2),
            "counters_per_step": counters,
            "image_finite": finite, "image_mean": round(mean, 6),
            # Contract form: algorithmic bytes / launch time against the HBM peak. The bytes are served by
            # L2 / Infinity Cache (fabric traffic is a tenth of them), so this fraction does not bind; the VALU
            # object below says what does.
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": prof_src,
                         "traffic_over_algorithmic": round(traffic / abytes, 4) if traffic else None,
                         "kernel": "render_kernel<COUNT=false, STACK, ROWS=8, WIDE=%s>" % ("true" if tree == 4 else "false"),
                         "algorithmic_bytes_per_launch": int(abytes),
                         "bytes_per_unit": {"inner_visit": 112 if tree == 4 else 64, "triangle_test": 40, "shaded_hit": 240, "pixel": 24}},
            # What binds: VALU issue under divergence. busy = wave-level VALU instructions x 4 cycles / (1024 SIMDs x
            # launch cycles); lanes = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); both from the profiled launch.
            "roofline_valu": {"bound": "valu", "busy": round(prof["valu_busy"], 4) if "valu_busy" in prof else None,
                              "lane_utilisation": round(prof["valu_lane_utilisation"], 4) if "valu_lane_utilisation" in prof else None,
                              "useful_issue_frac": round(prof["valu_useful_issue"], 4) if "valu_useful_issue" in prof else None,
                              "wave_instructions_per_launch": int(prof["valu_wave_instructions"]) if "valu_wave_instructions" in prof else None,
                              "peak": "1024 SIMDs x 1 wave64 VALU instruction / 4 cycles", "source": prof_src,
                              "live_lane_utilisation_instrumented": live_lanes},
        }
        if not args.no_cpu_baseline and world == 1:  # reported on rank 0 at N=1 only
            # every core this process may use: its affinity mask, and the share the GPU box assigns to one GPU's job
            # (it exports OMP_NUM_THREADS = 16 and kills jobs that oversubscribe; os.cpu_count() there is the whole
            # 8-GPU machine)
            allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            threads = max(1, min(allowed, int(os.environ.get("OMP_NUM_THREADS", str(allowed)))))
            out["cpu_baseline"] = cpu_baseline(scene, bvh, lst, threads)
            out["cpu_baseline"]["host_cpus"] = os.cpu_count()
        print(json.dumps(out), flush=True)

    r.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
