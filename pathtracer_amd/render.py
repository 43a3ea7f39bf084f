"""Batch loop and multi-GPU sample sharding — the role of src/main.cu:98-133, extended to N GPUs.

Single GPU: exactly the reference's loop (batch b uses seed + b, running mean).

N GPUs (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm): scene, BVH and
light table are replicated; batch b goes to rank b mod N (a batch is the smallest unit whose RNG
stream is reproducible without replay: seed + b, subsequence y*W+x). Each rank
keeps the running mean of ITS batches; at the end every rank scales its mean by
local_spp / total_spp and ONE sum-reduce over xGMI combines the float accumulators. No other
collective touches the data path.
"""
import torch
import torch.distributed as dist


def plan_batches(samples, samples_every_update, rank=0, world_size=1):
    """[(global_batch_index, seed_offset, current_samples)] for this rank, in launch order."""
    out = []
    done, b = 0, 0
    while done < samples:
        cur = min(samples_every_update, samples - done)
        if b % world_size == rank:
            out.append((b, b, cur))
        done += cur
        b += 1
    return out


def render_local(renderer, acc, seed, plan, stream=None):
    """Launches this rank's batches into `acc` (a float32 device tensor of H*W*3). Asynchronous.
    Returns the number of samples per pixel accumulated.

    Ordering rule: the launches go to `stream` (a non-zero hipStream_t handle, e.g. the .cuda_stream of a
    torch.cuda.Stream that is current while `acc` is produced and consumed) or, with stream=None, to the
    context's own non-blocking stream, which NO torch stream is ordered against: then call
    renderer.synchronize() before touching `acc` (combine() included)."""
    prev = 0
    for _, seed_offset, cur in plan:
        renderer.render_batch_into(acc.data_ptr(), seed + seed_offset, prev, cur, stream)
        prev += cur
    return prev


def combine(acc, local_spp, total_spp, group=None, dst=None):
    """Weighted sum of the per-rank running means. acc is modified in place and holds the final
    image on every rank (all_reduce) or on `dst` only (reduce)."""
    acc.mul_(float(local_spp) / float(total_spp))
    if dist.is_available() and dist.is_initialized():  # (a one-rank group still goes through the collective)
        if dst is None:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.reduce(acc, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return acc


def render_sharded(renderer, width, height, seed, samples, samples_every_update, device=None):
    """Full image on this rank's GPU, combined across ranks. Returns a (H, W, 3) float32 tensor."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    device = device if device is not None else torch.device("cuda", renderer.device)
    plan = plan_batches(samples, samples_every_update, rank, world)
    # One side stream carries everything that touches the accumulator: its zero fill, the render launches
    # (handed to the library by handle), the scaling and the reduce (torch orders NCCL work against the
    # current stream). The caller's stream waits for it at the end.
    caller = torch.cuda.current_stream(device)
    side = torch.cuda.Stream(device)
    side.wait_stream(caller)
    with torch.cuda.stream(side):
        acc = torch.zeros(height * width * 3, dtype=torch.float32, device=device)
        local = render_local(renderer, acc, seed, plan, side.cuda_stream)
        combine(acc, local, samples)
    caller.wait_stream(side)
    acc.record_stream(caller)
    return acc.view(height, width, 3)
