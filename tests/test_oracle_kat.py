"""Known-answer tests that pin the CPU oracle (oracle/ptx_oracle.c) -- no GPU."""
import os

import numpy as np
import pytest

F = np.float32


def test_pcg2d_kat_survey(orc):
    # derived from glsl:302-313,520 by integer arithmetic
    assert orc.seed_init(0, 0, 1) == (65536, 65536)
    seed = (65536, 65536)
    want = [((0xCC3F7525, 0x759E59CE), (0.79784328, 0.45944750)), ((0x7E1407F2, 0xC84F7A72), (0.49249315, 0.78246272)),
            ((0x0144E474, 0xF5D3A8DE), (0.0049574645, 0.96026093))]
    for st, fl in want:
        seed, out = orc.pcg2d(seed)
        assert seed == st
        assert out == pytest.approx(fl, rel=1e-7)
    assert orc.seed_init(10, 20, 12345) == (809041930, 809041940)
    seed, out = orc.pcg2d((809041930, 809041940))
    assert seed == (0x187094A6, 0x0A1EB633) and out == pytest.approx((0.095467843, 0.039531123), rel=1e-7)
    assert orc.seed_init(1919, 1079, 777) == (50923391, 50922551)
    assert orc.pcg2d((50923391, 50922551))[0] == (0x1AA41745, 0x44AC976A)


def test_pcg2d_matches_independent_numpy(orc):
    def ref(sx, sy):
        M = 0xFFFFFFFF
        sx = (1664525 * sx + 1013904223) & M; sy = (1664525 * sy + 1013904223) & M
        sx = (sx + 1664525 * sy) & M; sy = (sy + 1664525 * sx) & M
        sx ^= sx >> 16; sy ^= sy >> 16
        sx = (sx + 1664525 * sy) & M; sy = (sy + 1664525 * sx) & M
        sx ^= sx >> 16; sy ^= sy >> 16
        return sx, sy
    def cvt(u):     # the contract's vec2(uvec2): two roundings for words >= 2^31 (what SwiftShader does; ref_prim/*_rad_b1)
        return F(u) if u < 2**31 else F(F(u - 2**31) + F(2.0**31))
    rng = np.random.default_rng(1)
    differs = 0
    for sx, sy in rng.integers(0, 2**32, size=(400, 2)):
        seed, out = orc.pcg2d((int(sx), int(sy)))
        assert seed == ref(int(sx), int(sy))
        assert out[0] == float(F(cvt(seed[0]) * F(2.0**-32))) and out[1] == float(F(cvt(seed[1]) * F(2.0**-32)))
        differs += (cvt(seed[0]) != F(seed[0])) + (cvt(seed[1]) != F(seed[1]))
    assert 0 < differs < 80        # about 1 word in 32 is one ulp away from the single-rounding conversion
    # a word seen in the SwiftShader dump: RNE would give 0xEF2C3700, the two-step conversion gives 0xEF2C3800
    assert cvt(0xEF2C377D) == F(0xEF2C3800) and F(0xEF2C377D) == F(0xEF2C3700)


def test_rng_can_return_one(orc):
    # u32 -> f32 rounds up at the top of the range, so 1.0 is reachable
    assert float(F(F(0xFFFFFFFF) * F(2.0**-32))) == 1.0
    seed, out = orc.pcg2d((0, 0))
    assert 0.0 <= out[0] <= 1.0 and 0.0 <= out[1] <= 1.0


def test_seed_uses_low_16_timestamp_bits(orc):
    assert orc.seed_init(5, 9, 0x12345) == orc.seed_init(5, 9, 0x2345)
    assert orc.seed_init(5, 9, -1) == orc.seed_init(5, 9, 0xFFFF)


def test_sincos_contract_accuracy(orc):
    r = np.linspace(0, 1, 20001, dtype=np.float32)
    th = (F(2.0) * F(np.pi)) * r
    err = 0.0
    for t in th:
        s, c = orc.sincos_2pi(float(t))
        err = max(err, abs(s - np.sin(np.float64(t))), abs(c - np.cos(np.float64(t))))
    assert err < 7e-7          # the operation sequence SwiftShader evaluates (GLSL ES asks for 2^-11 absolute)
    assert orc.sincos_2pi(0.0) == (0.0, 1.0)


def test_sincos_sqrt_match_reference_implementation(orc):
    # float bits of sin / cos / sqrt evaluated by the GL implementation the reference frames come from, 16384 angles over
    # [0, 2 pi) (oracle/gen_ref_frames.py sincos -> ref_prim/probe_sincos.npz): the contract reproduces every bit
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_prim", "probe_sincos.npz"))
    assert "SwiftShader" in str(z["renderer"])
    th = z["theta_bits"].view(np.float32)
    idx = np.arange(th.size, dtype=np.float32)
    assert (((F(2.0) * F(np.pi)) * (idx * F(1.0 / th.size))).astype(np.float32).view(np.uint32) == z["theta_bits"]).all()
    got = np.array([orc.sincos_2pi(float(t)) for t in th], dtype=np.float32)
    assert (got[:, 0].view(np.uint32) == z["sin_bits"]).all() and (got[:, 1].view(np.uint32) == z["cos_bits"]).all()
    assert (np.sqrt((idx * F(1.0 / th.size)).astype(np.float32)).view(np.uint32) == z["sqrt_bits"]).all()      # sqrt: correctly rounded


def test_pow_matches_reference_implementation(orc):
    # ref_prim/probe_pow.npz: float bits of pow(x, 1/2.2) (the display pass, fragmentShaderOutput.glsl:35) for x = i / 16384
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_prim", "probe_pow.npz"))
    assert "SwiftShader" in str(z["renderer"])
    x = z["x_bits"].view(np.float32)
    assert (x == (np.arange(x.size, dtype=np.float32) * F(1.0 / x.size)).astype(np.float32)).all()
    y = float(F(1.0) / F(2.2))
    got = np.array([orc.pow_(float(v), y) for v in x], dtype=np.float32)
    assert (got.view(np.uint32) == z["pow_bits"]).all()
    m = x > 0
    assert np.abs(got[m] - x[m].astype(np.float64) ** (1 / 2.2)).max() < 3e-7


def test_min_max_nan_rule_matches_reference_implementation(orc):
    # ref_prim/probe_minmax.npz: min/max with NaN operands on that implementation return the SECOND operand
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_prim", "probe_minmax.npz"))
    val = {"qn": F(np.nan), "one": F(1.0), "ninf": F(-np.inf), "pinf": F(np.inf), "nan2": F(np.nan)}
    mn = lambda a, b: a if a < b else b      # noqa: E731  (the contract: oracle/ptx_oracle.c fmin_ / fmax_)
    mx = lambda a, b: a if a > b else b      # noqa: E731
    checked = 0
    for e, bits in zip(z["expr"], z["bits"]):
        e = str(e)
        if e[:4] in ("min(", "max("):
            a, b = [val[t.strip()] for t in e[4:-1].split(",")]
            got = (mn if e.startswith("min") else mx)(a, b)
            want = np.uint32(bits).view(np.float32)
            assert (np.isnan(got) and np.isnan(want)) or got == want, e
            checked += 1
    assert checked == 12


def test_triangle_epsilons(orc):
    # glsl:155: |a| < 1e-4 is rejected (absolute); a = e1 . (d x e2).  Triangle in the z=1 plane, ray along +z.
    def tri(scale):
        return [0, 0, 1], [scale, 0, 1], [0, scale, 1]
    o = [0.1 * 0.01, 0.1 * 0.01, 0]
    d = [0, 0, 1]
    big = tri(1.0)
    hit, t = orc.ray_tri(o, d, *big)
    assert hit and t == 1.0
    # a = scale^2 for this configuration: 0.0099^2 < 1e-4 -> invisible, 0.0101^2 > 1e-4 -> visible
    assert not orc.ray_tri([0.001, 0.001, 0], d, *tri(0.0099))[0]
    assert orc.ray_tri([0.001, 0.001, 0], d, *tri(0.0101))[0]
    # t must be > 1e-4 (glsl:168)
    assert not orc.ray_tri([0.1, 0.1, 1 - 0.00009], d, *big)[0]
    assert orc.ray_tri([0.1, 0.1, 1 - 0.00011], d, *big)[0]
    # behind the origin
    assert not orc.ray_tri([0.1, 0.1, 2], d, *big)[0]
    # two-sided
    assert orc.ray_tri([0.1, 0.1, 2], [0, 0, -1], *big)[0]
    # u + v == 1 edge is inside, beyond is outside; u,v = 0 edges inside
    assert orc.ray_tri([0.5, 0.5, 0], d, *big)[0]
    assert not orc.ray_tri([0.5, 0.5000001, 0], d, *big)[0]
    assert orc.ray_tri([0.0, 0.3, 0], d, *big)[0]
    assert not orc.ray_tri([-1e-7, 0.3, 0], d, *big)[0]


def test_box_slab_cases(orc):
    bmin, bmax = [0, 0, 0], [1, 1, 1]
    assert orc.ray_box([0.5, 0.5, -1], [0, 0, 1], bmin, bmax)          # axis-parallel: inf slabs
    assert not orc.ray_box([1.5, 0.5, -1], [0, 0, 1], bmin, bmax)
    assert not orc.ray_box([0.5, 0.5, 2], [0, 0, 1], bmin, bmax)       # box behind: tFar < 0
    assert orc.ray_box([0.5, 0.5, 0.5], [0, 0, 1], bmin, bmax)         # origin inside
    assert orc.ray_box([0.5, 0.5, 0.5], [0.577, 0.577, 0.577], bmin, bmax)
    # origin exactly on a bounding plane with a zero direction component: 0 * inf = NaN reaches min/max, which return
    # their SECOND operand (DESIGN.md "NaN slabs"; probe_minmax.npz).  On the min plane of x: t1 = NaN, t2 = +inf ->
    # tmin.x = +inf -> miss.  On the max plane of x: t1 = -inf, t2 = NaN -> tmin.x = tmax.x = NaN, which the outer
    # max/min drop as FIRST operands: the x slab is ignored -> hit.  The same in y hides x too; in z it rejects the box.
    assert not orc.ray_box([0.0, 0.5, -1], [0, 0, 1], bmin, bmax)
    assert orc.ray_box([1.0, 0.5, -1], [0, 0, 1], bmin, bmax)
    assert orc.ray_box([7.0, 1.0, -1], [0, 0, 1], bmin, bmax)              # y = max plane: x slab (a clear miss) is hidden
    assert not orc.ray_box([0.5, 0.5, 1.0], [1, 0, 0], bmin, bmax)         # z = max plane: NaN stays in tNear -> miss
    # flat box (a wall): tNear == tFar is a hit (<=)
    assert orc.ray_box([0.5, 0.5, -1], [0, 0, 1], [0, 0, 0.25], [1, 1, 0.25])


def test_hemisphere_sample_is_unit_and_above_surface(orc):
    rng = np.random.default_rng(3)
    for _ in range(300):
        n = rng.normal(size=3); n /= np.linalg.norm(n)
        r = rng.random(2)
        d = orc.sample_hemisphere(r, n.astype(np.float32))
        assert abs(np.linalg.norm(d.astype(np.float64)) - 1) < 1e-6
        assert np.dot(d, n) > -1e-6


def test_primary_ray_contract(orc, ptx):
    # glsl:501-514: u=(px+.5)/W; aspect>1: u=u*aspect-0.125; p=(LB+u*right)+v*up; d=normalize(p-src)
    p = ptx.config_params("pixar_room", 32, 18)
    dirs = orc.primary_dirs(p)
    aspect = F(32.0 / 18.0)
    for (px, py) in [(0, 0), (31, 17), (7, 11)]:
        u = F(F(F(px) + F(0.5)) / F(32)); v = F(F(F(py) + F(0.5)) / F(18))
        u = F(F(u * aspect) - F(0.125))
        ip = (p.cam_left_bottom + p.cam_right * u).astype(np.float32) + (p.cam_up * v).astype(np.float32)
        d = (ip - p.cam_src).astype(np.float32)
        dot = F(F(F(d[0] * d[0]) + F(d[1] * d[1])) + F(d[2] * d[2]))
        want = (d * F(F(1.0) / np.sqrt(dot))).astype(np.float32)
        assert (dirs[py, px].view(np.uint32) == want.view(np.uint32)).all()


def test_L0_scene_renders_black(orc, ptx):
    s = ptx.load_scene("cornell_faithful")
    p = ptx.config_params("cornell_faithful", 16, 16, 4, 4)
    rgb, c = orc.render_frame(s, p, 1000)
    assert s.L == 0 and not rgb.any() and c["light_attempts"] == 0


def test_counters_and_row_ranges(orc, ptx):
    s = ptx.load_scene("my_cornell_6")
    p = ptx.config_params("my_cornell_6", 24, 24, 3, 3)
    full, c = orc.render_frame(s, p, 1234)
    assert c["samples"] == 24 * 24 * 3 and c["rays"] >= c["samples"] and c["node_visits"] > c["rays"]
    part, _ = orc.render_rows(s, p, 1234, 0, 24, 5)
    for y in range(24):
        if y % 5 == 0:
            assert (part[y].view(np.uint32) == full[y].view(np.uint32)).all()
        else:
            assert not part[y].any()
    one, _ = orc.render_frame(s, p, 1234, nthreads=1)
    assert (one.view(np.uint32) == full.view(np.uint32)).all()  # thread count does not change results


def test_blend_matches_running_mean_for_square_images(orc, ptx):
    p = ptx.config_params("my_cornell_6", 8, 8)
    rng = np.random.default_rng(0)
    f0 = rng.random((8, 8, 3)).astype(np.float32); f1 = rng.random((8, 8, 3)).astype(np.float32)
    b0 = orc.blend_rgba8(p, f0, None, 0)
    assert (b0[..., :3] == np.floor(f0 * 255 + 0.5).astype(np.uint8)).all() and (b0[..., 3] == 255).all()
    b1 = orc.blend_rgba8(p, f1, b0, 1)
    want = (b0[..., :3].astype(np.float32) / 255 + f1) / 2
    assert np.abs(b1[..., :3].astype(np.float32) - want * 255).max() <= 0.51
