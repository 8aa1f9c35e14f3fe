// ptx_device.h -- device-side arithmetic of the path tracer (gfx950, wave64).
//
// Every function restates one piece of /root/reference/shaders/fragmentShaderPathTracing.glsl ("glsl:")
// as separately rounded IEEE binary32 operations.  The translation unit MUST be compiled with
// -ffp-contract=off (no FMA contraction) and without any fast-math flag; hipcc's default correctly
// rounded f32 divide / sqrt are relied upon.  Operation orders follow the arithmetic contract verified
// against the verbatim shader on SwiftShader:
//   dot(a,b) = (a.x*b.x + a.y*b.y) + a.z*b.z ; cross as written below ; normalize(v) = v * (1/sqrt(dot))
//   min/max = IEEE minNum/maxNum (v_min_f32 / v_max_f32) ; sin/cos = the fixed polynomial below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptx {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 mk(float4 v) { return mk(v.x, v.y, v.z); }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
__device__ __forceinline__ float length(V3 a) { return sqrtf(dot(a, a)); }
__device__ __forceinline__ V3 normalize(V3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
__device__ __forceinline__ float clamp01(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
__device__ __forceinline__ bool any_ne(V3 a, V3 b) { return a.x != b.x || a.y != b.y || a.z != b.z; }

#define PTX_M_PI 3.141592653589793238462643f   // glsl:2
#define PTX_M_1_PI 0.3183098861837907f         // glsl:3

// vec2(uvec2) of the contract: words >= 2^31 are converted as float(int(x - 2^31)) + 2^31f (two roundings), which is
// what the reference shader does on SwiftShader (GLSL leaves the rounding open; oracle/ptx_oracle.c u32_to_f32).
__device__ __forceinline__ float u32_to_f32(uint32_t x) {
    float f = (float)(int32_t)(x & 0x7FFFFFFFu);
    return (x >> 31) ? f + 2147483648.0f : f;
}

// glsl:302-313 (PCG2D).  Returns vec2(seed) * 2^-32.
__device__ __forceinline__ void pcg2d(uint32_t& sx, uint32_t& sy, float& r0, float& r1) {
    sx = 1664525u * sx + 1013904223u;
    sy = 1664525u * sy + 1013904223u;
    sx += 1664525u * sy;
    sy += 1664525u * sx;
    sx ^= sx >> 16;
    sy ^= sy >> 16;
    sx += 1664525u * sy;
    sy += 1664525u * sx;
    sx ^= sx >> 16;
    sy ^= sy >> 16;
    r0 = u32_to_f32(sx) * 2.32830643654e-10f;
    r1 = u32_to_f32(sy) * 2.32830643654e-10f;
}

__device__ __forceinline__ float rcp_det(float a);

// sincos contract (oracle/ptx_oracle.c ss_sin): the operation sequence SwiftShader evaluates for sin(x) --
// y = x / (2 pi) reduced to [-1/2, 1/2], sine / cosine polynomials of the quarter angle, two angle doublings, the second
// normalised by a correctly rounded 1 / (s2^2 + c2^2) -- and cos(x) = sin(x + pi/2); every op separately rounded.
__device__ __forceinline__ float ss_sin(float x) {
    float y = x * 1.59154943e-1f;
    y = y - __builtin_rintf(y);
    const float y2 = y * y;
    const float c1 = y2 * (y2 * (y2 * -0.0204391631f + 0.2536086171f) + -1.2336977925f) + 1.0f;
    const float s1 = y * (y2 * (y2 * (y2 * -0.0046075748f + 0.0796819754f) + -0.645963615f) + 1.5707963235f);
    const float c2 = c1 * c1 - s1 * s1;
    const float s2 = (2.0f * s1) * c1;
    return ((2.0f * s2) * c2) * rcp_det(s2 * s2 + c2 * c2);      // the denominator is 1 +- 1e-6: rcp_det == 1.0f / den
}
__device__ __forceinline__ void sincos_2pi(float theta, float& s, float& c) {
    s = ss_sin(theta);
    c = ss_sin(theta + 1.57079632f);
}

// pow(x, y) of the display pass (fragmentShaderOutput.glsl:35): the contract's operation sequence (oracle/ptx_oracle.c ss_log2,
// ss_exp2, orc_pow; pinned by tests/golden/ref_prim/probe_pow.npz): exp2(y * log2(x)), every op separately rounded.
__device__ __forceinline__ float ss_log2(float x) {
    const uint32_t xi = __float_as_uint(x);
    float x1 = __uint_as_float(((xi & 0x7F800000u) >> 8) | 0x3F800000u);
    x1 = (x1 - 1.4960938f) * 256.0f;
    const float m = __uint_as_float((xi & 0x007FFFFFu) | 0x3F800000u);
    float x2 = (9.5428179e-2f * m + 4.7779095e-1f) * m + 1.9782813e-1f;
    const float x3 = ((1.6618466e-2f * m + 2.0350508e-1f) * m + 2.7382900e-1f) * m + 4.0496687e-2f;
    x2 = x2 / x3;
    return x1 + (m - 1.0f) * x2;
}
__device__ __forceinline__ float ss_exp2(float x) {
    float x0 = x < __uint_as_float(0x43010000u) ? x : __uint_as_float(0x43010000u);          // min, max: second operand on NaN
    x0 = x0 > __uint_as_float(0xC2FDFFFFu) ? x0 : __uint_as_float(0xC2FDFFFFu);
    const int i = (int)__builtin_rintf(x0 - 0.5f);
    const float ii = __uint_as_float((uint32_t)(i + 127) << 23);
    const float f = x0 - (float)i;
    float ff = __uint_as_float(0x3AF61905u);
    ff = ff * f + __uint_as_float(0x3C134806u);
    ff = ff * f + __uint_as_float(0x3D64AA23u);
    ff = ff * f + __uint_as_float(0x3E75EAD4u);
    ff = ff * f + __uint_as_float(0x3F31727Bu);
    ff = ff * f + 1.0f;
    return ii * ff;
}
__device__ __forceinline__ float ss_pow(float x, float y) { return ss_exp2(y * ss_log2(x)); }

// glsl:315-337
__device__ __forceinline__ V3 sample_hemisphere(float r0, float r1, V3 n) {
    float theta = (2.0f * PTX_M_PI) * r0;
    float r = sqrtf(r1);
    float sn, cs;
    sincos_2pi(theta, sn, cs);
    float x = r * cs;
    float y = r * sn;
    float z = sqrtf(1.0f - r * r);
    V3 major = fabsf(n.x) < 0.57735026919f ? mk(1.f, 0.f, 0.f) : (fabsf(n.y) < 0.57735026919f ? mk(0.f, 1.f, 0.f) : mk(0.f, 0.f, 1.f));
    V3 u = normalize(cross(n, major));
    V3 v = cross(n, u);
    V3 dir = (u * x + v * y) + n * z;
    return normalize(dir);
}

// glsl:175-191 with the reciprocal hoisted out (inv = 1.0f / d, computed once per ray; same value).
__device__ __forceinline__ bool slab_test(V3 o, V3 inv, V3 bmin, V3 bmax, float& tNear) {
    V3 t1 = (bmin - o) * inv;
    V3 t2 = (bmax - o) * inv;
    float tn = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z));
    float tf = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z));
    tNear = tn;
    return (tn <= tf) && (tf > 0.0f);
}

// The same slab test on a packed node record (ptx_kernels.h): a = {min.x, min.y, max.x, max.y}, b = {min.z, max.z, ..}.
// The six subtractions and six multiplications are the same separately rounded f32 operations, issued as three
// v_pk_add_f32 + three v_pk_mul_f32 (two lanes of a register pair per instruction).
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bool slab_node(V3 o, V3 inv, float4 a, float4 b, float& tNear) {
    const f2 oxy = {o.x, o.y}, ixy = {inv.x, inv.y}, ozz = {o.z, o.z}, izz = {inv.z, inv.z};
    const f2 lo = (f2{a.x, a.y} - oxy) * ixy;     // t1.xy
    const f2 hi = (f2{a.z, a.w} - oxy) * ixy;     // t2.xy
    const f2 zz = (f2{b.x, b.y} - ozz) * izz;     // t1.z, t2.z
    float tn = fmaxf(fmaxf(fminf(lo.x, hi.x), fminf(lo.y, hi.y)), fminf(zz.x, zz.y));
    float tf = fminf(fminf(fmaxf(lo.x, hi.x), fmaxf(lo.y, hi.y)), fmaxf(zz.x, zz.y));
    tNear = tn;
    return (tn <= tf) && (tf > 0.0f);
}

// glsl:175-191 for a ray whose 1/d has an infinite (or NaN) component, on a record of the REFERENCE tree.  When such a ray
// also starts exactly in a bounding plane, 0 * inf = NaN reaches min/max, whose result GLSL leaves undefined; the contract
// follows SwiftShader, where min(a, b) = a < b ? a : b and max(a, b) = a > b ? a : b (a NaN operand yields b; probed with
// the derived debug shader).  A NaN in the y slab then hides the x and y slabs, one in the z slab rejects the box: the
// result depends on the reference's own hierarchy, so these rays walk the reference tree (DScene::ref_base).  tNear is NaN
// (then the box is a miss), the entry distance of the slabs that were not hidden (a lower bound), or -- a ray with TWO infinite
// components that starts in two bounding planes at once -- +inf with tFar = +inf as well, which the shader counts as a hit
// (inf <= inf): such a box must never be pruned, so the margin rule's limit is +inf for these rays (ray_lat = inf) and before the
// first hit (round 3; 3e38 until then, which lost one hit in 1.25e9 axis-parallel rays of ptx_selfcheck_pruning's family 6).
__device__ __forceinline__ float min_2nd(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float max_2nd(float a, float b) { return a > b ? a : b; }
__device__ __forceinline__ bool slab_node_ref(V3 o, V3 inv, float4 a, float4 b, float& tNear) {
    const f2 oxy = {o.x, o.y}, ixy = {inv.x, inv.y}, ozz = {o.z, o.z}, izz = {inv.z, inv.z};
    const f2 lo = (f2{a.x, a.y} - oxy) * ixy;     // t1.xy   (the same packed products as slab_node)
    const f2 hi = (f2{a.z, a.w} - oxy) * ixy;     // t2.xy
    const f2 zz = (f2{b.x, b.y} - ozz) * izz;     // t1.z, t2.z
    const float tn = max_2nd(max_2nd(min_2nd(lo.x, hi.x), min_2nd(lo.y, hi.y)), min_2nd(zz.x, zz.y));
    const float tf = min_2nd(min_2nd(max_2nd(lo.x, hi.x), max_2nd(lo.y, hi.y)), max_2nd(zz.x, zz.y));
    tNear = tn;
    return (tn <= tf) && (tf > 0.0f);
}
__device__ __forceinline__ bool inv_degenerate(V3 inv) {      // |x| < inf is false for inf and NaN
    return !(fabsf(inv.x) < __builtin_inff() && fabsf(inv.y) < __builtin_inff() && fabsf(inv.z) < __builtin_inff());
}

// The slab test on a record whose planes are stored entry-first for the ray's direction octant (leaf-SAH layouts):
// with 1/d finite and sign(1/d[k]) as the layout assumes, fl(near - o) * inv <= fl(far - o) * inv holds per axis (rounding is
// monotone), so min(t1, t2) IS the entry value and max(t1, t2) the exit value: the six selections of glsl:182-186 disappear.
// Only for rays whose 1/d has no infinite component (0 * inf = NaN would make the selection matter); others use slab_node.
__device__ __forceinline__ bool slab_node_ordered(V3 o, V3 inv, float4 a, float4 b, float& tNear) {
    const f2 oxy = {o.x, o.y}, ixy = {inv.x, inv.y}, ozz = {o.z, o.z}, izz = {inv.z, inv.z};
    const f2 en = (f2{a.x, a.y} - oxy) * ixy;     // entry x, y
    const f2 ex = (f2{a.z, a.w} - oxy) * ixy;     // exit x, y
    const f2 zz = (f2{b.x, b.y} - ozz) * izz;     // entry z, exit z
    const float tn = fmaxf(fmaxf(en.x, en.y), zz.x);
    const float tf = fminf(fminf(ex.x, ex.y), zz.y);
    tNear = tn;
    return (tn <= tf) && (tf > 0.0f);
}

// 1.0f / a for the Moeller-Trumbore determinant.  v_rcp_f32 followed by one Newton step in FMAs gives the correctly rounded
// quotient for EVERY binary32 a with 1e-4 <= |a| < 2^100 (checked exhaustively on gfx950: tools/dbg/rcpcheck.hip), in 3
// instructions instead of the 11 of the IEEE division expansion.  |a| < 1e-4 is rejected by the caller whatever f is
// (glsl:156-158); |a| >= 2^100 takes the true division.
__device__ __forceinline__ float rcp_det(float a) {
    if (__builtin_expect(fabsf(a) >= 0x1p100f, 0)) return 1.0f / a;
    const float r = __builtin_amdgcn_rcpf(a);
    return fmaf(fmaf(-a, r, 1.0f), r, r);
}

// glsl:150-173 with edge1 = v1 - v0 and edge2 = v2 - v0 precomputed (same f32 values).
// Straight-line form: every quantity is computed unconditionally and the shader's four early returns are folded
// into one predicate with the SAME comparisons (so NaN operands behave exactly as in the branchy original: a NaN u or
// v is not rejected by 'u < 0 || u > 1').  In SIMT the early returns save nothing unless all 64 lanes take them, and
// without them the three 16-B loads of a triangle record can all be issued up front.
__device__ __forceinline__ bool tri_test(V3 o, V3 d, V3 v0, V3 e1, V3 e2, float& t_out) {
    V3 h = cross(d, e2);
    float a = dot(e1, h);
    float f = rcp_det(a);
    V3 s = o - v0;
    float u = f * dot(s, h);
    V3 q = cross(s, e1);
    float v = f * dot(d, q);
    float t = f * dot(e2, q);
    t_out = t;
    return !(fabsf(a) < 0.0001f) && !(u < 0.0f || u > 1.0f) && !(v < 0.0f || u + v > 1.0f) && (t > 0.0001f);   // |a| < EPS == glsl:156 for every a, NaN included
}

// Two triangles A, B at once from a pair record (ptx_kernels.h "pairs"): every quantity is an f2 {A, B}, so the 48
// multiplications / additions of two Moeller-Trumbore tests issue as 24 packed instructions.  Operation order per
// triangle is exactly tri_test's; a padding B (all zeros) has a = 0 and is rejected by the determinant test.
__device__ __forceinline__ void tri_test2(V3 o, V3 d, float4 r0, float4 r1, float4 r2, float4 r3, float4 r4, bool& hitA, bool& hitB, float& tA, float& tB) {
    const f2 v0x = {r0.x, r0.y}, v0y = {r0.z, r0.w}, v0z = {r1.x, r1.y}, e1x = {r1.z, r1.w}, e1y = {r2.x, r2.y}, e1z = {r2.z, r2.w},
             e2x = {r3.x, r3.y}, e2y = {r3.z, r3.w}, e2z = {r4.x, r4.y};
    const f2 dx = {d.x, d.x}, dy = {d.y, d.y}, dz = {d.z, d.z};
    const f2 hx = dy * e2z - e2y * dz, hy = dz * e2x - e2z * dx, hz = dx * e2y - e2x * dy;          // cross(d, e2)
    const f2 a = (e1x * hx + e1y * hy) + e1z * hz;
    f2 f;                                                  // rcp_det for both, with ONE (wave-uniformly rare) range check
    if (__builtin_expect(fmaxf(fabsf(a.x), fabsf(a.y)) >= 0x1p100f, 0)) {
        f = f2{1.0f / a.x, 1.0f / a.y};
    } else {
        const f2 r = {__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)};
        f = f2{fmaf(fmaf(-a.x, r.x, 1.0f), r.x, r.x), fmaf(fmaf(-a.y, r.y, 1.0f), r.y, r.y)};
    }
    const f2 sx = f2{o.x, o.x} - v0x, sy = f2{o.y, o.y} - v0y, sz = f2{o.z, o.z} - v0z;
    const f2 u = f * ((sx * hx + sy * hy) + sz * hz);
    const f2 qx = sy * e1z - e1y * sz, qy = sz * e1x - e1z * sx, qz = sx * e1y - e1x * sy;          // cross(s, e1)
    const f2 v = f * ((dx * qx + dy * qy) + dz * qz);
    const f2 t = f * ((e2x * qx + e2y * qy) + e2z * qz);
    const f2 uv = u + v;
    tA = t.x; tB = t.y;
    // glsl:156 'a > -EPS && a < EPS' is '|a| < EPS' for every a, NaN included (all three comparisons are false then)
    hitA = !(fabsf(a.x) < 0.0001f) && !(u.x < 0.0f || u.x > 1.0f) && !(v.x < 0.0f || uv.x > 1.0f) && (t.x > 0.0001f);
    hitB = !(fabsf(a.y) < 0.0001f) && !(u.y < 0.0f || u.y > 1.0f) && !(v.y < 0.0f || uv.y > 1.0f) && (t.y > 0.0001f);
}

} // namespace ptx
