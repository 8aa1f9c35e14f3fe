"""The oracle pin: frames rendered by the REFERENCE shader itself (verbatim text, SwiftShader WebGL2; fixtures
written by oracle/gen_ref_frames.py) against the CPU oracle, and -- on the GPU box -- against the HIP path.

Bar: BYTE-IDENTICAL frames, every scene, every path length (1, 3, 4 and 8 bounces), single frames and the progressive
blend, once the two things GL leaves to the implementation outside the shader -- the float -> unorm8 store of the render
target and the texel NEAREST picks on an exact texel boundary -- follow SwiftShader's rules (oracle: blend store=1;
library: option compat_rgba8 = 2).  Inside the shader the contract already follows that implementation wherever GLSL
leaves room: uint -> float conversion, sin/cos, min/max of NaN (DESIGN.md "Arithmetic contract"; ref_prim/probe_*.npz).
Under the rules GL prefers for the store (round to nearest, floor: store=0 / compat_rgba8 = 1, and ptx_read_rgba8) a
handful of +-1 code values remain, and the boundary columns of wide progressive frames.
north_star tolerance for images: PSNR >= 40 dB against the reference (pixar_room, 8 bounces): met with infinite PSNR.
"""
import glob
import os

import numpy as np
import pytest

REF = os.path.join(os.path.dirname(__file__), "golden", "ref_frames")
CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(REF, "*.npz")))


def psnr(a, b):
    mse = ((a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64)) ** 2).mean()
    return np.inf if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def check(case, frames_rgba8, ref, swiftshader_rules=False):
    bounces = int(ref["max_path_length"])
    for f in range(ref["frames"].shape[0]):
        mine, want = frames_rgba8[f], ref["frames"][f]
        same = (mine == want).all(axis=2).mean()
        dmax = np.abs(mine.astype(np.int32) - want.astype(np.int32)).max()
        if swiftshader_rules:
            assert (mine == want).all(), (case, f, same, dmax)                                 # byte-identical
        elif "_prog_" in case:
            # preferred GL rules: +-1 code value per frame from the store rule, carried along by the blend; on wide images
            # the previous target is read at an aspect-shifted coordinate (glsl:501-512) that falls EXACTLY on a texel
            # boundary in some columns (16:9: every 9th), where SwiftShader's 16-bit coordinate lands one texel lower
            big = (np.abs(mine.astype(np.int32) - want.astype(np.int32)).max(axis=2) > 1)
            cols = np.unique(np.nonzero(big)[1])
            w = mine.shape[1]
            aspect = np.float32(w / mine.shape[0])
            on_boundary = [abs(((x + 0.5) / w * aspect - 0.125) * w - round(((x + 0.5) / w * aspect - 0.125) * w)) < 1e-3 for x in cols]
            assert same >= 0.90 and all(on_boundary) and (aspect > 1 or not big.any()), (case, f, same, cols)
        else:
            assert same >= 0.99 and dmax <= 1, (case, f, same, dmax, bounces)                  # store rule only: +-1 code value
        if case == "pixar_room_b8":
            assert psnr(mine, want) >= 40.0, psnr(mine, want)                                  # north_star tolerance


def load(ptx, case):
    ref = np.load(os.path.join(REF, case + ".npz"))
    name = str(ref["scene"])
    s = ptx.load_scene(name)
    p = ptx.config_params(name, int(ref["width"]), int(ref["height"]), int(ref["sample_count"]), int(ref["max_path_length"]))
    return ref, s, p


def test_fixtures_present():
    assert {"my_cornell_6_b1", "cornell_c1", "pixar_room_b8", "pixar_room_b1"} <= set(CASES)


@pytest.mark.parametrize("store", [0, 1])
@pytest.mark.parametrize("case", CASES)
def test_oracle_matches_reference_shader(ptx, orc, case, store):
    ref, s, p = load(ptx, case)
    assert "SwiftShader" in str(ref["renderer"])
    if "_prog_" in case:      # the reference's running blend: frameNumber = f, previous 8-bit target fed back (glsl:532-552)
        frames, prev = [], None
        for f, ts in enumerate(ref["timestamps"]):
            prev = orc.blend_rgba8(p, orc.render_frame(s, p, int(ts))[0], prev, f, store=store)
            frames.append(prev)
    else:
        frames = [orc.blend_rgba8(p, orc.render_frame(s, p, int(ts))[0], None, 0, store=store) for ts in ref["timestamps"]]
    check(case, frames, ref, swiftshader_rules=(store == 1))


DISP_CASES = [c for c in CASES if "_disp_" in c]


@pytest.mark.parametrize("case", DISP_CASES)
def test_oracle_display_pass_matches_reference_shader(orc, case):
    # `display`: the canvas after the reference's display pass (fragmentShaderOutput.glsl, pow(c, 1/2.2)) over each frame's
    # 8-bit target, before the row flip of main.js:464-473.  Byte-identical under SwiftShader's store rule; under the
    # preferred store rule at most one code value off.
    ref = np.load(os.path.join(REF, case + ".npz"))
    assert len(DISP_CASES) >= 3
    for f in range(ref["frames"].shape[0]):
        assert (orc.display_rgba8(ref["frames"][f], store=1) == ref["display"][f]).all()
        d = np.abs(orc.display_rgba8(ref["frames"][f], store=0).astype(np.int32) - ref["display"][f].astype(np.int32))
        assert d.max() <= 1 and (d == 0).mean() > 0.97


@pytest.mark.gpu
@pytest.mark.parametrize("rules", [1, 2])
@pytest.mark.parametrize("case", CASES)
def test_gpu_matches_reference_shader(ptx, orc, gpu_ctx, case, rules):
    ref, s, p = load(ptx, case)
    gpu_ctx.upload_scene(s)
    gpu_ctx.set_params(p)
    frames, shown = [], []                                  # shown: the display pass over the target (gamma, no flip)
    # option compat_rgba8: the library keeps the reference's two 8-bit targets itself (1 = preferred GL rules, 2 = SwiftShader's)
    gpu_ctx.set_option("compat_rgba8", rules)
    if "_prog_" in case:
        gpu_ctx.reset_accum()
        for f, ts in enumerate(ref["timestamps"]):
            gpu_ctx.render_frames([int(ts)], first_frame_number=f)
            frames.append(gpu_ctx.read_compat_rgba8())
            shown.append(gpu_ctx.read_compat_rgba8(apply_gamma=True, flip_y=False))
        gpu_ctx.reset_accum()                                                  # the same frames in one call (one batch)
        gpu_ctx.render_frames([int(ts) for ts in ref["timestamps"]])
        assert (gpu_ctx.read_compat_rgba8() == frames[-1]).all()
    else:
        for ts in ref["timestamps"]:
            gpu_ctx.reset_accum()
            gpu_ctx.render_frames([int(ts)])
            frames.append(gpu_ctx.read_compat_rgba8())                          # the reference's RGBA8 target at frameNumber 0
            shown.append(gpu_ctx.read_compat_rgba8(apply_gamma=True, flip_y=False))
            if rules == 1:
                assert (gpu_ctx.read_rgba8(apply_gamma=False, flip_y=False) == frames[-1]).all()
    gpu_ctx.set_option("compat_rgba8", 0)
    check(case, frames, ref, swiftshader_rules=(rules == 2))
    if "display" in ref.files:
        for f in range(len(shown)):
            if rules == 2:
                assert (shown[f] == ref["display"][f]).all(), (case, f)                        # byte-identical to the canvas
            # under either rule set: the display pass of the oracle over the library's own target (a +-1 difference of a
            # dark texel becomes up to 21 code values after the gamma curve, so the canvas itself is no bound for rules = 1)
            assert (shown[f] == orc.display_rgba8(frames[f], store=rules - 1)).all(), (case, f, rules)


# ---------------------------------------------------------------- primary-ray words from the derived debug shader (ref_prim/)
# Hit triangle id, bits of t and bits of the ray direction of every pixel, written by the reference's own ray generation,
# traversal and intersection code (oracle/gen_ref_frames.py derive_debug_shader adds only the word dump).  Integer / bit
# parity, so the bar is equality; pixels where SwiftShader itself disagrees would be listed as documented ties (none).
PRIM = os.path.join(os.path.dirname(__file__), "golden", "ref_prim")
PRIM_CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(PRIM, "*_prim.npz")))


def load_prim(ptx, case):
    z = np.load(os.path.join(PRIM, case + ".npz"))
    name = str(z["scene"])
    return z, ptx.load_scene(name), ptx.config_params(name, int(z["width"]), int(z["height"]))


def test_prim_fixtures_present():
    assert {"my_cornell_6_prim", "my_cornell_6_tall_prim", "pixar_room_prim", "teapot_prim"} <= set(PRIM_CASES)


@pytest.mark.parametrize("case", PRIM_CASES)
def test_oracle_primary_words_match_reference_shader(ptx, orc, case):
    z, s, p = load_prim(ptx, case)
    assert "SwiftShader" in str(z["renderer"])
    assert (orc.primary_dirs(p).view(np.uint32) == z["dir_bits"]).all()
    t, tri, _ = orc.primary_hits(s, p)
    assert (tri == z["tri"]).all() and (t.view(np.uint32) == z["t_bits"]).all()
    assert (z["tri"] >= 0).sum() > 0.3 * z["tri"].size


@pytest.mark.gpu
@pytest.mark.parametrize("case", PRIM_CASES)
def test_gpu_primary_words_match_reference_shader(ptx, gpu_ctx, case):
    z, s, p = load_prim(ptx, case)
    gpu_ctx.upload_scene(s)
    gpu_ctx.set_params(p)
    assert (gpu_ctx.primary_dirs().view(np.uint32) == z["dir_bits"]).all()
    for traversal in (0, 1, 2, 3):
        gpu_ctx.set_option("traversal", traversal)
        t, tri = gpu_ctx.primary_hits()
        assert (tri == z["tri"]).all() and (t.view(np.uint32) == z["t_bits"]).all(), traversal
    # the same rays through ptx_trace_closest with the default renderer's arithmetic (pair records, ordered slabs)
    d = gpu_ctx.primary_dirs().reshape(-1, 3)
    o = np.tile(np.asarray(p.cam_src, dtype=np.float32), (d.shape[0], 1))
    for pipeline in (0, 2):
        gpu_ctx.set_option("pipeline", pipeline)
        t, tri = gpu_ctx.trace_closest(o, d)
        assert (tri.reshape(z["tri"].shape) == z["tri"]).all() and (t.view(np.uint32).reshape(z["tri"].shape) == z["t_bits"]).all(), pipeline
    gpu_ctx.set_option("pipeline", 2)


# ---------------------------------------------------------------- radiance bits at maxPathLength = 1 (ref_prim/*_rad_b1)
# Float bits of clamp(mean radiance) (glsl:516-529) of one frame, dumped by the same derived shader.  With one bounce no
# value depends on sin/cos, so the WHOLE shading arithmetic (RNG words and their float conversion, light sampling, both
# visibility rules, solid angle, pdf, MIS weight, sample mean, clamp) is compared bit for bit: the bar is equality.
RAD_CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(PRIM, "*_rad_b1.npz")))


def load_rad(ptx, case):
    z = np.load(os.path.join(PRIM, case + ".npz"))
    name = str(z["scene"])
    return z, ptx.load_scene(name), ptx.config_params(name, int(z["width"]), int(z["height"]), int(z["sample_count"]), 1)


def test_rad_fixtures_present():
    assert {"my_cornell_6_rad_b1", "my_cornell_6_tall_rad_b1", "bedroom2_v5_rad_b1", "pixar_room_rad_b1"} <= set(RAD_CASES)


@pytest.mark.parametrize("case", RAD_CASES)
def test_oracle_radiance_bits_match_reference_shader(ptx, orc, case):
    z, s, p = load_rad(ptx, case)
    assert "SwiftShader" in str(z["renderer"])
    rgb, _ = orc.render_frame(s, p, int(z["timestamp"]))
    assert (rgb.view(np.uint32) == z["rgb_bits"]).all()
    assert (z["rgb_bits"].view(np.float32) > 0).mean() > 0.2          # lit pixels, not a black frame


@pytest.mark.gpu
@pytest.mark.parametrize("case", RAD_CASES)
def test_gpu_radiance_bits_match_reference_shader(ptx, gpu_ctx, case):
    z, s, p = load_rad(ptx, case)
    gpu_ctx.upload_scene(s)
    gpu_ctx.set_params(p)
    for pipeline in (0, 1, 2):
        gpu_ctx.set_option("pipeline", pipeline)
        gpu_ctx.reset_accum()
        gpu_ctx.render_frames([int(z["timestamp"])])
        assert (gpu_ctx.read_sum().view(np.uint32) == z["rgb_bits"]).all(), pipeline   # one frame: the sum IS the frame
    gpu_ctx.set_option("pipeline", 2)
