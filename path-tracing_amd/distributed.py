"""Multi-GPU partition of the path: one process per GPU, frames dealt round-robin, one all-reduce.

Every (pixel, frame) is independent and frames differ only in `timestamp`, so rank r of G renders frames
{r, r+G, ...} of the timestamp list over the FULL image into its own float accumulation buffer; the buffers are
summed by one all-reduce (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).  Splitting the
samples of one frame across GPUs would not be reference-equivalent (the clamp applies to the frame mean and the
seed chains through a pixel's samples, glsl:520-529) -- the split is by frame.
"""


def frames_for_rank(n_frames, rank, world):
    """Indices (into the timestamp list) of the frames rank `rank` renders."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    return list(range(rank, n_frames, world))


def allreduce_sum(tensor, group=None):
    """Sums the per-rank accumulation buffers in place; no-op for a single process."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=group)
    return tensor
