"""Multi-GPU behind the C ABI (include/ptx.h "multi-GPU"; reference seam: the frame loop of main.js:391-438
and the accumulation of modules/bufferManager.js:33-43 that the split + all-reduce replace).

What runs where: this box has ONE GPU and RCCL cannot put two ranks on one device, so
  * the multi-device context is exercised with the same device listed G times: every line of the routing code (scene /
    params fan-out, round-robin frame split, concurrent per-sub streams, reduction on read, counters, checkpoint restore)
    runs, with the sum taken by k_sum_accums instead of the grouped ncclAllReduce -- that one call is what stays untested here;
  * the one-process-per-GPU form runs through REAL RCCL with a world of one rank (ncclGetUniqueId, ncclCommInitRank,
    grouped ncclAllReduce of the float buffer and the frame count, ncclCommCount).
The N > 1 RCCL collective itself first runs in the driver's multi-GPU bench."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAME, W, H, SPP, B = "my_cornell_6", 80, 60, 3, 4


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def prepared(ptx, devices):
    ctx = ptx.Context(devices)
    ctx.upload_scene(ptx.load_scene(NAME))
    ctx.set_params(ptx.config_params(NAME, W, H, SPP, B))
    return ctx


def per_frame_images(ptx, ts):
    """clamp(mean) image of every frame, each rendered alone on a single-device context (0 + f == f exactly)."""
    out = []
    with prepared(ptx, 0) as ctx:
        for t in ts:
            ctx.reset_accum()
            ctx.render_frames([t])
            out.append(ctx.read_sum())
    return out


@pytest.mark.parametrize("G,n_frames", [(2, 6), (3, 7), (4, 2)])
def test_multi_device_context_equals_its_definition_bit_for_bit(ptx, G, n_frames):
    ts = ptx.scene.timestamps(n_frames)
    frames = per_frame_images(ptx, ts)
    # the definition: device g sums frames g, g + G, ... in order; the devices' sums are added in device order
    parts = []
    for g in range(G):
        acc = np.zeros((H, W, 3), np.float32)
        for f in range(g, n_frames, G):
            acc = acc + frames[f]
        parts.append(acc)
    want = parts[0]
    for g in range(1, G):
        want = want + parts[g]
    with prepared(ptx, [0] * G) as multi, prepared(ptx, 0) as single:
        assert multi.comm_world == G
        multi.render_frames(ts)
        assert multi.frames_done == n_frames
        got = multi.read_sum()
        assert (bits(got) == bits(want)).all()
        single.render_frames(ts)
        one = single.read_sum()
        assert np.abs(got - one).max() <= 4 * np.finfo(np.float32).eps * n_frames      # float-add association only
        assert (bits(multi.read_linear()) == bits(want / np.float32(n_frames))).all()
        # the display read-back sees the same total
        ref8 = prepared(ptx, 0)
        ref8.write_sum(want, n_frames)
        assert (multi.read_rgba8() == ref8.read_rgba8()).all()
        ref8.close()
        # counters are summed over the devices
        cm, cs = multi.counters(), single.counters()
        for k in ("samples", "rays", "rays_traced", "ray_hits", "light_attempts", "pixel_frames"):
            assert cm[k] == cs[k], k
        # a second call continues the progressive sum; reading in between does not disturb it
        more = ptx.scene.timestamps(3, first=n_frames)
        multi.render_frames(more, first_frame_number=n_frames)
        single.render_frames(more, first_frame_number=n_frames)
        assert multi.frames_done == n_frames + 3
        assert np.abs(multi.read_sum() - single.read_sum()).max() <= 4 * np.finfo(np.float32).eps * (n_frames + 3)
        # primary hits come from device 0 and equal the single-device ones
        assert (multi.primary_hits()[1] == single.primary_hits()[1]).all()


def test_multi_device_checkpoint_restore_and_refusals(ptx):
    ts = ptx.scene.timestamps(4)
    with prepared(ptx, [0, 0]) as multi, prepared(ptx, 0) as single:
        single.render_frames(ts[:2])
        saved = single.read_sum()
        multi.write_sum(saved, 2)                       # resume from a checkpoint
        assert multi.frames_done == 2
        multi.render_frames(ts[2:], first_frame_number=2)
        single.render_frames(ts[2:], first_frame_number=2)
        assert multi.frames_done == 4
        assert np.abs(multi.read_sum() - single.read_sum()).max() <= 4 * np.finfo(np.float32).eps * 4
        multi.reset_accum()
        assert multi.frames_done == 0 and float(np.abs(multi.read_sum()).max()) == 0.0
        with pytest.raises(ptx.PtxError):
            multi.set_option("compat_rgba8", 1)
        with pytest.raises(ptx.PtxError):
            multi.comm_init(b"\0" * 128, 0, 1)
        with pytest.raises(ptx.PtxError):
            multi.allreduce()
        multi.render_frames_noalloc(ts[:2])             # the no-alloc entry point routes the same way
        single.reset_accum()
        single.render_frames(ts[:2])
        assert np.abs(multi.read_sum() - single.read_sum()).max() <= 4 * np.finfo(np.float32).eps * 2
    with pytest.raises(ptx.PtxError) as e:
        ptx.Context([0, 0, 1])                          # neither all distinct nor all the same
    assert "distinct" in str(e.value)


def test_library_rccl_communicator_world_of_one(ptx):
    import torch
    uid = ptx.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ts = ptx.scene.timestamps(3)
    with prepared(ptx, 0) as ctx:
        assert ctx.comm_world == 0
        with pytest.raises(ptx.PtxError):
            ctx.allreduce()                             # no communicator yet
        ctx.comm_init(uid, 0, 1)
        assert ctx.comm_world == 1                      # ncclCommCount
        ctx.render_frames(ts)
        before = ctx.read_sum()
        buf = torch.full((H, W, 4), -1.0, dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()
        assert ctx.allreduce(buf.data_ptr()) == 3       # into a caller buffer: own accumulation untouched
        assert (bits(buf[..., :3].cpu().numpy()) == bits(before)).all()
        assert ctx.allreduce() == 3 and ctx.frames_done == 3      # in place
        assert (bits(ctx.read_sum()) == bits(before)).all()
        with pytest.raises(ptx.PtxError):
            ctx.comm_init(uid, 1, 1)                    # rank out of range


def test_read_after_two_unsynchronised_calls_on_a_large_frame(ptx):
    """Every sub alternates two launch slots (streams); the reduction on read must wait for BOTH slots of EVERY sub.  Two
    render_frames calls without a read in between on a frame large enough that sub 0's second call (slot 1) is still running
    when the host reaches the read; equals the definition bit for bit (round-2 review: multi_reduce only waited for slot 0)."""
    name, w, h, spp, b = "pixar_room", 1920, 1080, 2, 4
    ts = ptx.scene.timestamps(4)

    def ctx_for(devices):
        c = ptx.Context(devices)
        c.upload_scene(ptx.load_scene(name))
        c.set_params(ptx.config_params(name, w, h, spp, b))
        return c

    frames = []
    with ctx_for(0) as one:
        for t in ts:
            one.reset_accum()
            one.render_frames([t])
            frames.append(one.read_sum())
    with ctx_for([0, 0]) as multi:
        for rep in range(3):                         # repeated: the hole was a timing matter
            multi.reset_accum()
            multi.render_frames(ts[:2])              # frame 0 -> sub 0 (slot 0), frame 1 -> sub 1 (slot 0)
            multi.render_frames(ts[2:], first_frame_number=2)   # frame 2 -> sub 0 (slot 1), frame 3 -> sub 1 (slot 1)
            got = multi.read_sum()
            want = (frames[0] + frames[2]) + (frames[1] + frames[3])
            assert (bits(got) == bits(want)).all()
        # one-frame calls all land on sub 0 and alternate its two slots
        multi.reset_accum()
        for k, t in enumerate(ts):
            multi.render_frames([t], first_frame_number=k)
        got = multi.read_sum()
        want = ((frames[0] + frames[1]) + frames[2]) + frames[3]
        assert (bits(got) == bits(want)).all()
