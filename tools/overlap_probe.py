#!/usr/bin/env python3
"""What would overlapping consecutive batches buy? Two contexts on one GPU hold the same scene; the job's batches
are issued alternately to them, each on its own stream, so that the second batch's tiles fill the wave slots the first
batch's tail leaves empty. Wall time of 2k batches issued that way against 2k batches on one stream.
  python3 tools/overlap_probe.py [--scene many_lights|dragon] [--pairs 2]"""
import argparse, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pathtracer_amd import core, scenes
from tests import util

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="many_lights")
ap.add_argument("--pairs", type=int, default=2)
ap.add_argument("--res", type=int, default=1024)
ap.add_argument("--spp", type=int, default=50)
a = ap.parse_args()
with tempfile.TemporaryDirectory() as d:
    b = util.load_bundle(scenes.write_many_lights_standin(d)) if a.scene == "many_lights" else util.golden_scene(a.scene)
rs = []
for k in range(2):
    r = core.Renderer(0)
    r.upload(b.scene, b.bvh, b.lst)
    r.configure(a.res, a.res)
    rs.append(r)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
imgs = [torch.zeros(a.res * a.res * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
torch.cuda.synchronize()


def run(which):  # which[i] = context of batch i
    for k in range(2):  # warm-up: tree choice, tile order
        rs[k].render_batch_into(imgs[k].data_ptr(), 7, 0, a.spp, streams[k].cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i, k in enumerate(which):
        rs[k].render_batch_into(imgs[k].data_ptr(), 42 + i, 0, a.spp, streams[k].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


n = 2 * a.pairs
for rep in range(3):
    serial = run([0] * n)
    both = run([i % 2 for i in range(n)])
    print(f"{a.scene} {a.res}^2 x {a.spp} spp, {n} batches: one stream {serial:.2f} ms ({serial / n:.2f} per batch), "
          f"two streams {both:.2f} ms ({both / n:.2f} per batch)", flush=True)
for r in rs:
    r.close()
