"""ptx_render_frames_noalloc (the entry point that allocates nothing: static __device__ scratch) against
ptx_render_frames, through the C ABI -- bit-exact, on two calls in a row, the second with a smaller input."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def render(ctx, ptx, name, w, h, spp, bounces, ts, noalloc):
    ctx.upload_scene(ptx.load_scene(name))
    ctx.set_params(ptx.config_params(name, w, h, spp, bounces))
    ctx.reset_accum()
    (ctx.render_frames_noalloc if noalloc else ctx.render_frames)(ts)
    ctx.synchronize()
    return ctx.read_sum(), ctx.frames_done


def test_noalloc_matches_render_frames_twice(ptx):
    # own contexts: this test must pass alone in a fresh process and after the other tests
    big = ("pixar_room", 160, 90, 4, 6, ptx.scene.timestamps(70))         # 70 frames: more than one 64-frame batch
    small = ("my_cornell_6", 40, 64, 3, 4, ptx.scene.timestamps(3, first=5))
    with ptx.Context(0) as ref, ptx.Context(0) as ctx:
        for case in (big, small):                                          # two calls in a row, the second smaller
            want, n_want = render(ref, ptx, *case, noalloc=False)
            got, n_got = render(ctx, ptx, *case, noalloc=True)
            assert n_got == n_want == len(case[5])
            assert want.max() > 0
            assert (bits(got) == bits(want)).all()
        # all pipelines share the scratch path
        for pipeline in (0, 1, 2):
            ref.set_option("pipeline", pipeline)
            ctx.set_option("pipeline", pipeline)
            want, _ = render(ref, ptx, *small, noalloc=False)
            got, _ = render(ctx, ptx, *small, noalloc=True)
            assert (bits(got) == bits(want)).all()


def test_noalloc_rejects_images_beyond_the_scratch(ptx):
    with ptx.Context(0) as ctx:
        ctx.upload_scene(ptx.load_scene("cornell"))
        ctx.set_params(ptx.config_params("cornell", 2049, 2048, 1, 1))     # 4 196 352 pixels > 4 194 304
        with pytest.raises(ptx.PtxError) as e:
            ctx.render_frames_noalloc([1000])
        assert e.value.status == ptx.binding.PTX_ERR_INVALID and "static scratch" in str(e.value)
        ctx.set_params(ptx.config_params("cornell", 2048, 2048, 1, 1))     # exactly the limit: one frame per batch
        ctx.render_frames_noalloc([1000, 1001])
        ctx.synchronize()
        assert ctx.frames_done == 2


def test_noalloc_scratch_has_one_owner_at_a_time(ptx):
    # two contexts on one GPU share that GPU's static scratch: the second caller is refused until the first has synchronised
    ts = ptx.scene.timestamps(2)
    with ptx.Context(0) as a, ptx.Context(0) as b:
        for c in (a, b):
            c.upload_scene(ptx.load_scene("my_cornell_6"))
            c.set_params(ptx.config_params("my_cornell_6", 64, 48, 2, 3))
        a.render_frames_noalloc(ts)                                         # queued, not synchronised: a owns the scratch
        with pytest.raises(ptx.PtxError) as e:
            b.render_frames_noalloc(ts)
        assert e.value.status == ptx.binding.PTX_ERR_STATE and "in use by another context" in str(e.value)
        a.synchronize()
        b.render_frames_noalloc(ts)                                         # released by the synchronisation
        assert (bits(a.read_sum()) == bits(b.read_sum())).all()
        a.render_frames_noalloc(ts)                                         # b's read-back synchronised b: free again
        a.synchronize()


def test_noalloc_leaves_free_device_memory_unchanged(ptx):
    import torch
    ts = ptx.scene.timestamps(3)
    with ptx.Context(0) as ctx:
        ctx.upload_scene(ptx.load_scene("my_cornell_6"))
        ctx.set_params(ptx.config_params("my_cornell_6", 256, 256, 2, 3))
        ctx.render_frames_noalloc(ts)                                       # warm-up: primary-hit cache, occupancy query, code objects
        ctx.synchronize()
        free0, _ = torch.cuda.mem_get_info(0)
        ctx.render_frames_noalloc(ts)
        free1, _ = torch.cuda.mem_get_info(0)                               # while the work is still queued / running
        ctx.synchronize()
        free2, _ = torch.cuda.mem_get_info(0)
        assert free0 == free1 == free2
