"""world_size-2 gloo test of the multi-GPU partition (frames round-robin + one all-reduce) -- CPU only.
The CPU oracle stands in for the per-rank renderer; what is under test is frames_for_rank + allreduce_sum."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_frames, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ptx_import
    m = ptx_import.load()
    from oracle import orc
    s = m.load_scene("my_cornell_6")
    p = m.config_params("my_cornell_6", 16, 16, 2, 3)
    ts = m.scene.timestamps(n_frames)
    mine = m.frames_for_rank(n_frames, rank, world)
    acc, _ = orc.render_frames(s, p, ts[mine], nthreads=1)
    t = torch.from_numpy(acc.copy())
    m.allreduce_sum(t)
    if rank == 0:
        np.save(os.path.join(out_dir, "sum.npy"), t.numpy())
    dist.destroy_process_group()


def test_frames_for_rank_partition(ptx):
    for n in (0, 1, 5, 64):
        for world in (1, 2, 3, 8):
            parts = [ptx.frames_for_rank(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(q) for q in parts) - min(len(q) for q in parts) <= 1


def test_two_rank_frame_split_equals_single_process(ptx, orc, tmp_path):
    n_frames, world = 5, 2
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "sum.npy")
    s = ptx.load_scene("my_cornell_6")
    p = ptx.config_params("my_cornell_6", 16, 16, 2, 3)
    want, _ = orc.render_frames(s, p, ptx.scene.timestamps(n_frames), nthreads=1)
    # same frames, different float-add association: a few ulp of the sum (values in [0, n_frames])
    assert np.abs(got - want).max() <= 4 * np.finfo(np.float32).eps * n_frames


def _gpu_worker(rank, world, port, n_frames, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)      # both ranks share cuda:0; the buffer is staged through the host
    import ptx_import
    m = ptx_import.load()
    s = m.load_scene("my_cornell_6")
    p = m.config_params("my_cornell_6", 40, 24, 3, 4)
    ctx = m.Context(0)
    ctx.upload_scene(s); ctx.set_params(p)
    seen = []
    total = m.render_frames_distributed(ctx, m.scene.timestamps(n_frames), batch=2, on_batch=lambda buf, n: seen.append((n, float(buf[..., :3].sum()))))
    np.save(os.path.join(out_dir, f"sum{rank}.npy"), ctx.read_sum())
    np.save(os.path.join(out_dir, f"seen{rank}.npy"), np.asarray(seen, dtype=np.float64))
    assert total == n_frames and ctx.frames_done == n_frames
    ctx.close()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.gpu
def test_render_frames_distributed_two_ranks_one_gpu(ptx, gpu_ctx, tmp_path):
    """render_frames_distributed: 5 frames over 2 ranks in progress batches of 2 == one process rendering all 5."""
    n_frames, world = 5, 2
    port = 29500 + (os.getpid() + 7) % 2000
    mp.spawn(_gpu_worker, args=(world, port, n_frames, str(tmp_path)), nprocs=world, join=True)
    s = ptx.load_scene("my_cornell_6")
    p = ptx.config_params("my_cornell_6", 40, 24, 3, 4)
    gpu_ctx.upload_scene(s); gpu_ctx.set_params(p); gpu_ctx.reset_accum()
    gpu_ctx.render_frames(ptx.scene.timestamps(n_frames))
    want = gpu_ctx.read_sum()
    a, b = np.load(tmp_path / "sum0.npy"), np.load(tmp_path / "sum1.npy")
    assert (a.view(np.uint32) == b.view(np.uint32)).all()                      # every rank ends with the same buffer
    assert np.abs(a - want).max() <= 4 * np.finfo(np.float32).eps * n_frames   # float-add association only
    seen = np.load(tmp_path / "seen0.npy")
    assert [int(x) for x in seen[:, 0]] == [4, 5] and seen[0, 1] < seen[1, 1]  # previews: 2 + 2 frames, then all 5


def _comm_fail_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if rank == 0:
        os.environ["PTX_RCCL_LIB"] = "/nonexistent/librccl.so"      # rank 0 cannot load RCCL; rank 1 may or may not (no GPU here)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ptx_import
    m = ptx_import.load()

    class NoCtx:                                                     # never reached: the id exchange fails first
        def comm_init(self, *a):
            raise AssertionError("ptx_comm_init must not be called when a rank has no RCCL")
    msg = "no error"
    try:
        m.distributed.init_library_comm(NoCtx())
    except m.PtxError as e:
        msg = str(e)
    # the collective that bench.py runs next must still line up on both ranks
    flag = torch.tensor([rank + 1], dtype=torch.int32)
    dist.all_reduce(flag)
    open(os.path.join(out_dir, f"r{rank}.txt"), "w").write(f"{int(flag.item())}|{msg}")
    dist.destroy_process_group()


def test_init_library_comm_fails_on_every_rank_together(tmp_path):
    """rank 0 raising before the broadcast left the other ranks in it.  Now every rank raises the same error."""
    port = 29500 + (os.getpid() + 13) % 2000
    mp.spawn(_comm_fail_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (open(tmp_path / f"r{k}.txt").read().split("|", 1) for k in (0, 1))
    assert r0[0] == r1[0] == "3"                                     # the following all-reduce matched up
    assert "rank 0" in r0[1] and "PTX_RCCL_LIB" in r0[1]
    assert r0[1] == r1[1]
