"""Multi-GPU partition of the path: one process per GPU, frames dealt round-robin, one all-reduce.

Every (pixel, frame) is independent and frames differ only in `timestamp`, so rank r of G renders frames
{r, r+G, ...} of the timestamp list over the FULL image into its own float accumulation buffer; the buffers are
summed by one all-reduce: the library's own RCCL communicator when the context has one (ptx_comm_init / ptx_allreduce,
include/ptx.h; init_library_comm below hands the id round over torch.distributed), else torch.distributed on the exported
buffer (backend "nccl" = RCCL over xGMI; "gloo", staged through the host, in CPU tests and one-GPU rehearsals).  Splitting the
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


def init_library_comm(ctx, group=None):
    """Creates the library's RCCL communicator on `ctx` for the ranks of the torch.distributed group: rank 0 draws the id
    (ptx_comm_unique_id), the group hands its 128 bytes round, every rank calls ptx_comm_init.  Returns the rank count RCCL
    reports.  Collective: every rank of the group must call it, and every rank gets the same outcome -- if librccl cannot
    be loaded or the id cannot be drawn on ANY rank, all ranks raise the same PtxError after the exchange (nobody is left
    waiting in a broadcast or inside ncclCommInitRank), and a failure inside ptx_comm_init on one rank is raised on all."""
    import torch.distributed as dist
    from . import binding
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    # every rank proves that it can load librccl and draw an id (cheap); rank 0's id is the one that is used
    try:
        mine = (binding.comm_unique_id(), "")
    except Exception as e:                                  # noqa: BLE001 - travels to the other ranks
        mine = (None, f"rank {rank}: {e}")
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    bad = [why for uid, why in everyone if uid is None]
    if bad:
        raise binding.PtxError(binding.PTX_ERR_HIP, "init_library_comm: no RCCL communicator on any rank because " + "; ".join(bad))
    err = ""
    try:
        ctx.comm_init(everyone[0][0], rank, world)
    except Exception as e:                                  # noqa: BLE001
        err = f"rank {rank}: {e}"
    errs = [None] * world
    dist.all_gather_object(errs, err, group=group)
    errs = [e for e in errs if e]
    if errs:
        raise binding.PtxError(binding.PTX_ERR_HIP, "init_library_comm: ptx_comm_init failed: " + "; ".join(errs))
    return ctx.comm_world


def render_frames_distributed(ctx, timestamps, batch=None, on_batch=None, group=None):
    """Renders `timestamps` (the whole job's frame list, identical on every rank) split by frame over the ranks of the
    process group, on this rank's `Context` ctx, and leaves the SUM OVER ALL RANKS in every rank's accumulation buffer
    (ctx.read_sum() / read_linear() / read_rgba8() then give the same image everywhere).  ctx must start from a reset
    accumulation.  Counterpart of the reference's progressive loop (main.js:391-436) for G GPUs:

      batch    : frames each rank renders between two progress points (None: all of its frames in one go)
      on_batch : callback(sum_tensor, frames_done_all_ranks) after every batch with the all-reduced running sum, a
                 (H, W, 4) float32 tensor on ctx's device -- a preview or checkpoint hook; the per-rank accumulation is
                 not touched by it, so previews cost one extra all-reduce each and change no result.

    One all-reduce of W*H*4 floats per progress point and one at the end; with backend "nccl" it runs over RCCL on the
    device buffer, with any other backend (gloo in rehearsals) the buffer is staged through the host.
    Returns the total number of frames accumulated."""
    import numpy as np
    import torch
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    world = dist.get_world_size(group) if multi else 1
    rank = dist.get_rank(group) if multi else 0
    ts = np.ascontiguousarray(timestamps, dtype=np.int32)
    mine = frames_for_rank(len(ts), rank, world)
    dev = torch.device("cuda", ctx.device)
    # torch.empty: nothing of torch's is queued on this buffer, so the library's stream can write it at once (a zero-fill
    # on torch's stream could land after a copy made on the context's own stream)
    buf = torch.empty(ctx.params.height, ctx.params.width, 4, dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    on_device = multi and dist.get_backend(group) == "nccl"
    library_comm = multi and ctx.comm_world == world          # ptx_comm_init was called for this group

    def reduce_into_buf():
        if library_comm:                                       # one grouped ncclAllReduce inside the library, into buf
            return ctx.allreduce(buf.data_ptr())
        ctx.accum_export_device(buf.data_ptr())
        ctx.synchronize()
        done = torch.tensor([ctx.frames_done], dtype=torch.int64, device=dev if on_device else "cpu")
        if multi:
            if on_device:
                dist.all_reduce(buf, group=group)
            else:
                h = buf.cpu()
                dist.all_reduce(h, group=group)
                buf.copy_(h)
            dist.all_reduce(done, group=group)
            torch.cuda.synchronize(dev)
        return int(done.item())

    # every rank goes through the same number of progress points, also when the frames do not divide evenly
    most = (len(ts) + world - 1) // world                      # frames of the busiest rank
    step = most if not batch else max(1, int(batch))
    rounds = 1 if step <= 0 else max(1, -(-most // step))
    total = 0
    for r in range(rounds):
        part = mine[r * step:(r + 1) * step] if step > 0 else []
        if len(part):
            ctx.render_frames(ts[part], first_frame_number=int(part[0]))
        if on_batch is not None or r == rounds - 1:
            total = reduce_into_buf()
            if on_batch is not None:
                on_batch(buf, total)
    ctx.accum_import_device(buf.data_ptr(), total)
    ctx.synchronize()
    return total
