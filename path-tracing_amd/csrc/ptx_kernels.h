// ptx_kernels.h -- device data layout + kernels of the progressive path tracer (gfx950).
//
// HBM layout (replaces the eleven RGB32F data textures of main.js:272-282; DESIGN.md "Data layout"):
//   nodes[2N] float4: [2i]   = { min.x, min.y, max.x, max.y }                            (one 32-B record per node; the
//                     [2i+1] = { min.z, max.z, bits(miss link), fc }                      pairs feed v_pk_*_f32)
//                     fc >= 0: leaf, first | count << 24.  fc < 0: inner node (or empty leaf): sign bit | the node's two
//                     position-error coefficients for provable pruning as 15 + 16 bit floats (prune_coefs below); -2 = none (NaN)
//   tris[3T]  float4: [3k]   = { v0.xyz, bits(original triangle index) }  -- indexed by INORDER SLOT k, so a
//                     [3k+1] = { (v1 - v0).xyz, packed coefs of the leaf at its first slot }   leaf's triangles are contiguous
//                     [3k+2] = { (v2 - v0).xyz, 0 }                           (one 48-B record per triangle)
//   pairs[5T] float4: pair p of the leaf that starts at slot `first` is record first + p (80 B) and holds slots
//                     first + 2p (A) and first + 2p + 1 (B, zeros when the leaf's count is odd), component-interleaved:
//                     {v0x A,B, v0y A,B} {v0z A,B, e1x A,B} {e1y A,B, e1z A,B} {e2x A,B, e2y A,B}
//                     {e2z A,B, bits(triangles of the leaf from this pair on), packed coefs of the leaf (first pair only)}   (pipeline 2 only)
//   attrN/attrC/attrE[T] float4 { normal | color | emission } by ORIGINAL triangle index
//   light[L]  6 x float4 { v0, v1, v2, normal, emission, {1/(L*area),0,0,0} }
// One node visit = 32 B, one triangle test = 48 B, attributes (48 B) only for the winning hit.
#pragma once
#include "ptx_device.h"

namespace ptx {

#ifndef PTX_MAX_CLS
#define PTX_MAX_CLS 8
#endif
#ifndef PTX_MAX_LBOX
#define PTX_MAX_LBOX 4      // boxes around the emissive triangles (light_bound): any-hit termination of light-visibility rays
#endif
#define PTX_CLS_DELTA 2.0e-3f   // a triangle belongs to a class when its unit normal is within this distance of +- the class normal
#define PTX_CLS_WCAP 512.0f     // the class bound is derived for w_ray <= this (rays steeper than 1/512 to every class plane)

struct DScene {
    const float4* nodes;   // 2 float4 per node
    const float4* tris;    // 3 float4 per inorder slot
    const float4* pairs;   // 5 float4 per triangle pair, indexed by first slot of the leaf + pair number
    const float4* attrN;
    const float4* attrC;
    const float4* attrE;
    const float4* light;   // 6 float4 per light
    int N, T, L;
    uint32_t oct_stride;   // bytes between the 8 direction-octant layouts of `nodes` (0: a single layout)
    uint32_t ref_base;     // byte offset in `nodes` of the reference tree itself (layout 8 behind the octant layouts; 0 when
                           // `nodes` IS the reference tree): rays with an infinite 1/d component walk it (slab_node_ref)
    V3 center;             // centre of the scene's bounding box                         } provable pruning: ray_kappa()
    float lat;              // margin rule: lateral allowance, option "margin_lat" (ppm) x the scene's diagonal (ray_lat)
    float rs, e_max, m_cap; // its half-diagonal, the scene's longest triangle edge, the largest M_ray the coefficients hold for }
    int n_cls;             // dominant plane normals of the scene's large flat triangles ("classes", unit vectors)
    V3 cls[PTX_MAX_CLS];
    // any-hit termination of light-visibility rays (pipeline 2; 0 boxes = off): boxes around ALL triangles with a non-zero
    // emission (padded for the margin rule), and the largest pruning coefficients among those triangles
    uint32_t n_lbox, lcoefs;
    float lbox[PTX_MAX_LBOX][6];   // min.xyz, max.xyz
    // provable rule only (PRUNE = 1), scenes with at most PTX_MAX_EMIT emissive triangles: per emitter {(e2 x e1) / n, 0} with
    // n = 1.01 * 90 u E^2 (light_bound): the determinant the shader's triangle test will compute for THIS ray is bounded from
    // below per emitter, instead of by the shader's 1e-4 cut-off for all of them.  emit_c1 = 0.672 / min_k E_k
    const float4* emit;
    uint32_t n_emit;
    float emit_c1;
};
#define PTX_MAX_EMIT 32

struct DParams {
    uint32_t W, H;
    V3 src, up, right, lb;
    float aspect;
    int max_len, spp;
};

// device-side work counters (u64), one block of these per context
enum { CNT_SAMPLES = 0, CNT_RAYS, CNT_RAYS_TRACED, CNT_NODE_VISITS, CNT_TRI_TESTS, CNT_RAY_HITS, CNT_LIGHT_ATTEMPTS, CNT_PIXEL_FRAMES,
       CNT_WALK_USEFUL, CNT_WALK_SLOTS, CNT_TRI_USEFUL, CNT_TRI_SLOTS, CNT_TRACE_USEFUL, CNT_TRACE_SLOTS, CNT_CYC_WALK, CNT_CYC_TRI, CNT_CYC_SHADE, CNT_IT_WALK, CNT_IT_TRI, CNT_IT_SHADE, CNT_NUM };

struct LocalCnt {
    uint32_t samples, rays, rays_traced, node_visits, tri_tests, ray_hits, light_attempts;
    // SIMT-efficiency diagnostics (COUNT builds only): lane-iterations that did work vs 64 x wave-iterations
    uint32_t walk_useful, walk_slots, tri_useful, tri_slots, trace_useful, trace_slots;
};

// max of v over the ACTIVE lanes (usable inside divergent code: built from ballots only)
__device__ __forceinline__ uint32_t active_max(uint32_t v) {
    bool cand = true;
    uint32_t mx = 0;
#pragma unroll
    for (int b = 15; b >= 0; b--) {
        const bool bit = (v >> b) & 1u;
        if (__ballot(cand && bit)) { mx |= 1u << b; cand = cand && bit; }
    }
    return mx;
}
__device__ __forceinline__ bool first_active_lane() {
    return (__ffsll((unsigned long long)__ballot(1)) - 1) == (int)(threadIdx.x & 63);
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ void flush_counters(const LocalCnt& c, unsigned long long* g, uint32_t pixel_frames) {
    uint32_t s0 = wave_sum(c.samples), s1 = wave_sum(c.rays), s2 = wave_sum(c.rays_traced), s3 = wave_sum(c.node_visits),
             s4 = wave_sum(c.tri_tests), s5 = wave_sum(c.ray_hits), s6 = wave_sum(c.light_attempts), s7 = wave_sum(pixel_frames);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g[CNT_SAMPLES], (unsigned long long)s0);
        atomicAdd(&g[CNT_RAYS], (unsigned long long)s1);
        atomicAdd(&g[CNT_RAYS_TRACED], (unsigned long long)s2);
        if (s3) atomicAdd(&g[CNT_NODE_VISITS], (unsigned long long)s3);
        if (s4) atomicAdd(&g[CNT_TRI_TESTS], (unsigned long long)s4);
        atomicAdd(&g[CNT_RAY_HITS], (unsigned long long)s5);
        atomicAdd(&g[CNT_LIGHT_ATTEMPTS], (unsigned long long)s6);
        atomicAdd(&g[CNT_PIXEL_FRAMES], (unsigned long long)s7);
    }
    if (c.walk_slots | c.tri_slots | c.trace_slots | c.walk_useful | c.tri_useful | c.trace_useful) {
        atomicAdd(&g[CNT_WALK_USEFUL], (unsigned long long)c.walk_useful); atomicAdd(&g[CNT_WALK_SLOTS], (unsigned long long)c.walk_slots);
        atomicAdd(&g[CNT_TRI_USEFUL], (unsigned long long)c.tri_useful); atomicAdd(&g[CNT_TRI_SLOTS], (unsigned long long)c.tri_slots);
        atomicAdd(&g[CNT_TRACE_USEFUL], (unsigned long long)c.trace_useful); atomicAdd(&g[CNT_TRACE_SLOTS], (unsigned long long)c.trace_slots);
    }
}

// ---- best-t pruning (option "traversal" bit 0).  The reference never prunes (glsl:207-249), so a node may only be skipped
// when NO triangle beneath it can produce a computed t below the best one found so far.  Template parameter PRUNE:
//   0  no pruning
//   1  PROVABLE pruning (option "prune" = 1; DESIGN.md "Pruning and exactness").  For a hit the triangle test accepts, the point
//      o + t_computed * d lies within rho of the triangle, hence of every box around it, where (rounding-error analysis of
//      glsl:150-173; host: prune_coefs)
//          rho <= c_o * M_ray                      for ANY triangle: the accepted determinant is >= 1e-4 in magnitude (glsl:156)
//          rho <= c_c * M_ray * w_ray              for a triangle of a CLASS (its normal within PTX_CLS_DELTA of one of the
//                                                  scene's n_cls dominant plane normals N_j): there the determinant is
//                                                  >= 2 * area * (|d . N_j| - delta), so w_ray = max_j 1 / (|d . N_j| - delta)
//      with c_o, c_c depending only on the triangle's edges and M_ray = max(|o - centre| + Rs, longest edge) bounding the distances involved (ray_m).
//      A point within rho of a box is inside the box grown by rho, which the ray enters no later than (slab entry of the
//      box) - rho * max|1/d_k|.  So with kappa = M_ray * max|1/d_k| and kappa2 = kappa * w_ray (ray_kappa, once per ray), a
//      node whose entry distance tn satisfies
//          tn - (c_o(node) * kappa + c_c(node) * kappa2)  >  best * (1 + 2^-16)
//      cannot hold a computed t <= best, and skipping it cannot change the (t, slot) minimum.  c_o(node) / c_c(node) are the
//      largest coefficients beneath the node (every triangle counts in exactly one of the two): inner nodes carry them in
//      fc (see the layout above), leaves in their first triangle / pair record.  Rays with an infinite 1/d component, rays
//      that are not unit length and rays from far outside the scene get kappa = inf and are never pruned.
//   2  heuristic margin  tn > best * (1 + 1e-3) + 1e-3 + lat  (option "prune" = 2, the library default): faster, but NOT proven
//      for every conceivable ray (include/ptx.h "prune").
#define PTX_PRUNE_REL 0.001f
#define PTX_PRUNE_ABS 0.001f
// lat, round 3.  The shader's triangle test accepts grazing hits (determinant near its 1e-4 cut-off) whose point lies BESIDE the
// triangle -- 3e-5 .. 2e-4 scene units beside a lamp panel of pixar_room (1e-5 of the scene's diagonal) in the six pixel-frames of
// 2.1e9 that round 2's margin got wrong on C5's whole job (tools/whole_job_check.py, tools/dbg/find_diff.py) -- and a ray with a
// small direction component enters the triangle's box later by that distance times max|1/d_k|: 0.017 .. 0.055 in those rays,
// against a margin of 0.004 .. 0.009.  So the margin grows by lat = margin_lat x scene diagonal x max|1/d_k| (ray_lat; a ray with
// a component below 1e-3 is hardly pruned at all).  Option "margin_lat" in ppm of the diagonal, default 3000 = 300 x the largest
// distance seen: ptx_selfcheck_pruning's adversarial rays, which break round 2's rule (0 ppm) 1 700 times in 5e8, find nothing
// from 3000 ppm on in 5e8 (100 / 300 / 1000 ppm: 16 / 12 / 2, all on teapot; in 7.5e10: 3 .. 8 per 1.25e9 on teapot, rays fired from
// hundreds of units outside the scene, none elsewhere); headline 58.7 -> 59.7 ms per frame (100 ppm: 58.9).
// Still a heuristic: the proven bound on that distance is c_o * M_ray (PRUNE = 1).
#define PTX_MARGIN_LAT_PPM 3000
#define PTX_PRUNE_SAFE_REL 1.52587890625e-05f   // 2^-16: covers the three roundings of a computed slab entry distance

// lat (margin rule only): the ray's lateral allowance in units of t, ray_lat below
template <int PRUNE>
__device__ __forceinline__ float prune_limit(float best, float lat = 0.0f) {
    return PRUNE == 2 ? best * (1.0f + PTX_PRUNE_REL) + PTX_PRUNE_ABS + lat : best * (1.0f + PTX_PRUNE_SAFE_REL);
}
// the two coefficients of a node, packed by the host (pack_coefs): [sign][c_o: 8-bit exponent, 7-bit mantissa][c_c: 0, 8, 7],
// both rounded up
__device__ __forceinline__ float coef_other(uint32_t packed) { return __uint_as_float(packed & 0x7FFF0000u); }
__device__ __forceinline__ float coef_class(uint32_t packed) { return __uint_as_float(packed << 16); }
struct Kappa { float k, k2; };
// is a node (entry distance tn, packed coefficients) out of reach of the ray (limit from prune_limit, kappa from ray_kappa)?
// NaN / inf operands (coefficients of an empty leaf, kappa of an unprunable ray) make the comparison false: not pruned.
template <int PRUNE>
__device__ __forceinline__ bool prune_far(float tn, float limit, uint32_t packed, Kappa kap) {
    return PRUNE == 2 ? tn > limit : (tn - (coef_other(packed) * kap.k + coef_class(packed) * kap.k2)) > limit;
}
// kappa and kappa2 of a ray, a hair above their exact values (v_rcp_f32 is good to 1 ulp); inf when the bound does not apply
// M of the lemma for a ray from o: any M >= max(|o - v0|, E) over the scene's triangles.  Every vertex lies within Rs of the
// centre, so |o - v0| <= |o - centre| + Rs; every edge is at most e_max.  (v_sqrt_f32 is good to 1 ulp, the squared length to 4u.)
__device__ __forceinline__ float ray_m(const DScene& S, V3 o) {
    const V3 v = o - S.center;
    return fmaxf(__builtin_amdgcn_sqrtf(dot(v, v)) * 1.0001f + S.rs, S.e_max);
}
__device__ __forceinline__ Kappa ray_kappa(const DScene& S, V3 o, V3 d) {
    const float m = ray_m(S, o);
    const float mind = fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
    const float dd = dot(d, d);
    const bool ok = fabsf(dd - 1.0f) <= 1.0e-3f && m <= S.m_cap;          // false for NaN
    float minc = 1.0f;                                                     // min_j |d . N_j| (1: no classes, w = 1 / (1 - delta'))
    if (S.n_cls > 0) {                                                     // the host fills the unused entries with copies of the first: constant indices,
#pragma unroll                                                             // no dependent scalar load per class
        for (int j = 0; j < PTX_MAX_CLS; j++) minc = fminf(minc, fabsf(dot(d, S.cls[j])));
    }
    const float den = minc - (1.002f * PTX_CLS_DELTA + 1.0e-5f);          // also covers |d| up to 1.0005 and the dot's rounding
    Kappa r;
    r.k = ok ? (m * __builtin_amdgcn_rcpf(mind)) * 1.001f : __builtin_inff();   // mind = 0 -> inf
    r.k2 = den >= (1.0f / PTX_CLS_WCAP) ? (r.k * __builtin_amdgcn_rcpf(den)) * 1.001f : __builtin_inff();   // w_ray <= WCAP, else unprunable
    return r;
}
// kappa pair in one 32-bit word (LDS of pipeline 2): two 16-bit floats, rounded up (inf stays inf, NaN stays NaN)
__device__ __forceinline__ uint32_t pack_kappa(Kappa k) {
    const uint32_t a = __float_as_uint(k.k), b = __float_as_uint(k.k2);
    const uint32_t ha = (a >> 16) + ((a & 0xFFFFu) ? 1u : 0u), hb = (b >> 16) + ((b & 0xFFFFu) ? 1u : 0u);
    return (ha << 16) | (hb & 0xFFFFu);
}
__device__ __forceinline__ Kappa unpack_kappa(uint32_t w) { Kappa k; k.k = __uint_as_float(w & 0xFFFF0000u); k.k2 = __uint_as_float(w << 16); return k; }

// margin rule: lateral allowance of a ray with reciprocal direction inv (inf or NaN: the ray is never pruned)
__device__ __forceinline__ float ray_lat(const DScene& S, V3 inv) { return S.lat * fmaxf(fmaxf(fabsf(inv.x), fabsf(inv.y)), fabsf(inv.z)); }
// the same from the direction (pipeline 2's TRI lanes hold the owner's d, not 1/d): v_rcp_f32 of the smallest component
__device__ __forceinline__ float ray_lat_d(const DScene& S, V3 d) { return S.lat * __builtin_amdgcn_rcpf(fminf(fminf(fabsf(d.x), fabsf(d.y)), fabsf(d.z))); }

// Light-visibility rays (glsl:353-363, 439-442) consume only "is there a hit" and the emission of the closest one.  A hit on
// a triangle WITHOUT emission settles both -- hit, not visible -- as soon as no emissive triangle can be hit nearer; the
// closest hit itself is then another triangle without emission or that one, and every such triangle has emission (0,0,0),
// so the shader's result is the same whichever of them is reported.  light_bound returns a distance tl such that
//     prune_far<PRUNE>(tl, prune_limit<PRUNE>(t_b), S.lcoefs, kappa)
// means "no emissive triangle yields a computed t <= t_b" -- the pruning rule applied to the boxes around the emissive
// triangles instead of BVH nodes (the rule only needs a box that contains the triangle).  A box may be left out when the
// ray cannot hit anything in it: PRUNE = 2, boxes padded on upload: the slab test misses it; PRUNE = 1 (provable): it lies
// wholly behind the ray even after growing it by the position-error bound (t <= tFar + slack and t > 0 for any hit).
// PRUNE = 1 with an emitter list (S.n_emit > 0): the slack is computed for the ray at hand.  The lemma (DESIGN.md "Pruning and
// exactness") bounds the distance of an accepted hit's computed point from its triangle by rho = (u / |a^|) 90 E^2 M + 20 u M,
// where a^ is the determinant tri_test computes, provided 61 u M E <= |a^|.  a^ is within 7 u E^2 of d . (e2 x e1).  The host
// stores G' = fl((e2 x e1) / n) with n = 1.01 * 90 u E^2, so r = |fl(d . G')| is |a^| / n to within (7 + 3.1 + 1) u E^2 / n = 0.122
// (rounding of the dot product and of G' included); with c0 = 0.1761 (16 u E^2 / n), r - c0 <= |a^| / n.  The ray's coefficient
// per unit of M is max_k n_k / |a^_k| + 1.01 * 20 u <= 1 / (min_k r_k - c0) + 1.21e-6, its slack that times kappa
// (M max|1/d_k|), as for the per-node coefficients; the lemma's premise 61 u M E_k <= |a^_k| holds for every emitter when
// min_k r_k - c0 >= M * 0.672 / min_k E_k (61 / (1.01 * 90) = 0.671).  A ray that grazes an emitter fails that and never ends early.
template <int PRUNE>
__device__ __forceinline__ float light_bound(const DScene& S, V3 o, V3 d, V3 inv, Kappa kap) {
    float tl = 3.0e38f;
    float slack = PRUNE == 1 ? coef_other(S.lcoefs) * kap.k + coef_class(S.lcoefs) * kap.k2 : 0.0f;
    const bool listed = PRUNE == 1 && S.n_emit != 0u;
    if (listed) {
        const float M = ray_m(S, o) * 1.0001f;
        float rmin = 3.0e38f;
        for (uint32_t k = 0; k < S.n_emit; k += 4) {                       // the host pads the list to a multiple of 4: four loads in flight
            const float4 g0 = S.emit[k], g1 = S.emit[k + 1], g2 = S.emit[k + 2], g3 = S.emit[k + 3];   // (eight: register spills, 103 against 97 ms)
            rmin = fminf(rmin, fminf(fminf(fabsf(dot(d, mk(g0))), fabsf(dot(d, mk(g1)))), fminf(fabsf(dot(d, mk(g2))), fabsf(dot(d, mk(g3))))));
        }
        rmin = rmin - 0.1761f;
        const bool ok = rmin >= M * S.emit_c1;                             // false for NaN
        slack = ok ? ((__builtin_amdgcn_rcpf(rmin) * 1.001f + 1.21e-6f) * kap.k) * 1.001f : __builtin_inff();
    }
    for (uint32_t k = 0; k < S.n_lbox; k++) {
        const float* b = S.lbox[k];
        const float ax = (b[0] - o.x) * inv.x, bx = (b[3] - o.x) * inv.x;
        const float ay = (b[1] - o.y) * inv.y, by = (b[4] - o.y) * inv.y;
        const float az = (b[2] - o.z) * inv.z, bz = (b[5] - o.z) * inv.z;
        const float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
        const float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
        const bool out = PRUNE == 2 ? (tn > tf || tf <= 0.0f) : (tf * (1.0f - PTX_PRUNE_SAFE_REL) + slack < 0.0f);   // false for NaN
        if (!out) tl = fminf(tl, tn);
    }
    return listed ? tl - slack : tl;      // the listed slack is already in the distance (light_settles)
}
// does a blocker at distance tb settle a light-visibility ray whose light_bound is tl?  (Margin rule: without the lateral allowance
// of prune_limit -- the emitter boxes are padded sideways instead, by 5e-4 of the scene's diagonal, ptx_hip.hip "lbox".)
template <int PRUNE>
__device__ __forceinline__ bool light_settles(const DScene& S, float tl, float tb, Kappa kap) {
    if (PRUNE == 1 && S.n_emit != 0u) return tl > prune_limit<1>(tb);     // false for NaN and -inf
    return prune_far<PRUNE>(tl, prune_limit<PRUNE>(tb), S.lcoefs, kap);
}

// 32-bit byte offsets from a uniform base: lets the compiler use the SGPR-base + VGPR-offset addressing form.
__device__ __forceinline__ float4 ld16(const float4* base, uint32_t byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// direction octant of a ray: bit k set = d[k] < 0 (selects the node layout whose child order is front-to-back)
__device__ __forceinline__ uint32_t octant(V3 d) { return (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u); }

// Leaves whose box was hit are collected PTX_LEAF_BATCH at a time before their triangles are tested, so the
// lanes of a wave spend their time in two tight loops (node walk / triangle tests) instead of waiting on each
// other's leaf loops inside the walk.  Legal because the threaded walk never depends on triangle results
// (the reference does not prune); with PRUNE the cut uses the best t of earlier batches only (still conservative).
#ifndef PTX_LEAF_BATCH
#define PTX_LEAF_BATCH 4
#endif

// glsl:207-249: closest hit by the stackless threaded traversal (hit -> i+1, miss -> missLink).
// Triangles are still tested in increasing inorder slot, and strict '<' keeps the first candidate among equal t
// (the reference's tie rule).  PRUNE adds the conservative best-t cut; COUNT adds node/triangle counters.
template <int PRUNE, bool COUNT>
__device__ __forceinline__ bool closest_hit(const DScene& S, V3 o, V3 d, float& out_t, int& out_tri, LocalCnt& c) {
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    Kappa kappa = {0.f, 0.f};
    if (PRUNE == 1) kappa = ray_kappa(S, o, d);
    float best = 1.0e38f;
    int best_tri = -1;
    uint32_t best_k = 0xFFFFFFFFu;        // inorder slot of the winner: ties on t go to the smaller slot (reference order)
    uint32_t i = 0;                       // -1 (end) wraps to 0xFFFFFFFF >= N
    const uint32_t N = (uint32_t)S.N;
    float limit = __builtin_inff();       // no hit yet: nothing is out of reach (a box the reference enters "at +inf", below, included)
    const float lat = PRUNE == 2 ? ray_lat(S, inv) : 0.0f;
    const bool deg = inv_degenerate(inv);
    const uint32_t nbase = deg ? S.ref_base : octant(d) * S.oct_stride;
    do {
        int pend[PTX_LEAF_BATCH];
        float pend_tn[PTX_LEAF_BATCH];
        int n = 0;
        uint32_t it_walk = 0, it_tri = 0;
        while (n < PTX_LEAF_BATCH && i < N) {
            if (COUNT) it_walk++;
            const float4 a = ld16(S.nodes, nbase + (i << 5));
            const float4 b = ld16(S.nodes, nbase + (i << 5) + 16);
            if (COUNT) c.node_visits++;
            float tn;
            bool hit = deg ? slab_node_ref(o, inv, a, b, tn) : slab_node(o, inv, a, b, tn);
            const int fc = __float_as_int(b.w);
            if (PRUNE == 2) hit = hit && !(tn > limit);
            if (PRUNE == 1 && fc < 0) hit = hit && !prune_far<1>(tn, limit, (uint32_t)fc, kappa);   // leaves: checked with their own coefficients below
            if (hit) {
                if (fc >= 0) {
#pragma unroll
                    for (int q = 0; q < PTX_LEAF_BATCH; q++) if (q == n) { pend[q] = fc; pend_tn[q] = tn; }
                    n++;
                }
                i++;
            } else {
                i = (uint32_t)__float_as_int(b.z);
            }
        }
        int j = 0;
        uint32_t k = 0, kend = 0;
        for (;;) {
            if (k == kend) {
                if (j == n) break;
                int fc = pend[0];
                float ltn = pend_tn[0];
#pragma unroll
                for (int q = 1; q < PTX_LEAF_BATCH; q++) if (q == j) { fc = pend[q]; ltn = pend_tn[q]; }
                j++;
                k = (uint32_t)(fc & 0x00FFFFFF);
                kend = k + (uint32_t)((fc >> 24) & 0x7F);
                if (k == kend) continue;  // empty leaf
                if (PRUNE == 1 && prune_far<1>(ltn, limit, __float_as_uint(ld16(S.tris, k * 48u + 16u).w), kappa)) { k = kend; continue; }
            }
            if (COUNT) it_tri++;
            const float4 v0 = ld16(S.tris, k * 48u);
            const float4 e1 = ld16(S.tris, k * 48u + 16u);
            const float4 e2 = ld16(S.tris, k * 48u + 32u);
            if (COUNT) c.tri_tests++;
            float t;
            if (tri_test(o, d, mk(v0), mk(e1), mk(e2), t) && (t < best || (t == best && k < best_k))) {
                best = t;
                best_k = k;
                best_tri = __float_as_int(v0.w);
                if (PRUNE) limit = prune_limit<PRUNE>(best, lat);
            }
            k++;
        }
        if (COUNT) {
            const uint32_t mw = active_max(it_walk), mt = active_max(it_tri);
            c.walk_useful += it_walk; c.tri_useful += it_tri;
            if (first_active_lane()) { c.walk_slots += 64u * mw; c.tri_slots += 64u * mt; }
        }
    } while (i < N);
    out_t = best;
    out_tri = best_tri;
    return best < 1.0e38f;
}

// The same query with pipeline 2's arithmetic building blocks, one ray per lane and nothing deferred: entry-first slab test
// on the octant layouts (slab_node_ordered, rays with an infinite 1/d component excepted) and two triangles per step from
// the pair records (tri_test2).  ptx_trace_closest uses it when pipeline 2 is selected, so known-answer rays around the
// shader's epsilons go through exactly the code the default renderer runs.
template <int PRUNE, bool COUNT>
__device__ __forceinline__ bool closest_hit_pairs(const DScene& S, V3 o, V3 d, float& out_t, int& out_tri, LocalCnt& c) {
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    Kappa kappa = {0.f, 0.f};
    if (PRUNE == 1) kappa = ray_kappa(S, o, d);
    const bool deg = inv_degenerate(inv);
    const bool ordered = S.oct_stride != 0u && !deg;
    float best = 1.0e38f, limit = __builtin_inff();
    const float lat = PRUNE == 2 ? ray_lat(S, inv) : 0.0f;
    uint32_t best_k = 0xFFFFFFFFu;
    uint32_t i = 0;
    const uint32_t N = (uint32_t)S.N;
    const uint32_t nbase = deg ? S.ref_base : octant(d) * S.oct_stride;
    while (i < N) {
        const float4 a = ld16(S.nodes, nbase + (i << 5));
        const float4 b = ld16(S.nodes, nbase + (i << 5) + 16);
        if (COUNT) c.node_visits++;
        float tn;
        bool hit = ordered ? slab_node_ordered(o, inv, a, b, tn) : (deg ? slab_node_ref(o, inv, a, b, tn) : slab_node(o, inv, a, b, tn));
        const int fc = __float_as_int(b.w);
        if (PRUNE == 2) hit = hit && !(tn > limit);
        if (PRUNE == 1 && fc < 0) hit = hit && !prune_far<1>(tn, limit, (uint32_t)fc, kappa);
        if (!hit) { i = (uint32_t)__float_as_int(b.z); continue; }
        i++;
        if (fc < 0) continue;                                  // inner node (or empty leaf)
        uint32_t k = (uint32_t)fc & 0x00FFFFFFu, pbyte = k * 80u;
        for (bool first_pair = true;; first_pair = false) {
            const float4 r0 = ld16(S.pairs, pbyte), r1 = ld16(S.pairs, pbyte + 16u), r2 = ld16(S.pairs, pbyte + 32u),
                         r3 = ld16(S.pairs, pbyte + 48u), r4 = ld16(S.pairs, pbyte + 64u);
            if (PRUNE == 1 && first_pair && prune_far<1>(tn, limit, __float_as_uint(r4.w), kappa)) break;   // the leaf's own coefficients
            const int rem = __float_as_int(r4.z) & 0xFF;      // bits 16, 17: emission flags of A and B (pipeline 2's any-hit rule)
            bool hA, hB;
            float tA, tB;
            tri_test2(o, d, r0, r1, r2, r3, r4, hA, hB, tA, tB);
            if (COUNT) c.tri_tests += rem >= 2 ? 2u : 1u;
            if (hA && (tA < best || (tA == best && k < best_k))) { best = tA; best_k = k; }
            if (hB && (tB < best || (tB == best && k + 1u < best_k))) { best = tB; best_k = k + 1u; }
            if (rem <= 2) break;
            k += 2u; pbyte += 80u;
        }
        if (PRUNE) limit = prune_limit<PRUNE>(best, lat);
    }
    out_t = best;
    out_tri = best < 1.0e38f ? __float_as_int(ld16(S.tris, best_k * 48u).w) : -1;
    return best < 1.0e38f;
}

// glsl:501-514 + 145-148: primary ray direction of pixel (px, py), GL origin (bottom-left).
__device__ __forceinline__ V3 primary_dir(const DParams& P, uint32_t px, uint32_t py) {
    float u = ((float)px + 0.5f) / (float)P.W;
    float v = ((float)py + 0.5f) / (float)P.H;
    if (P.aspect > 1.0f) { u = u * P.aspect; u = u - 0.125f; }
    else if (P.aspect < 1.0f) { v = v / P.aspect; v = v - 0.125f; }
    V3 ip = (P.lb + P.right * u) + P.up * v;
    return normalize(ip - P.src);
}

// 8x8 pixel tiles: one wave = one tile (coherent primary rays, 64 contiguous-ish pixels).
__device__ __forceinline__ bool tile_pixel(const DParams& P, uint32_t tile, uint32_t lane, uint32_t& px, uint32_t& py) {
    const uint32_t tiles_x = (P.W + 7) >> 3;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    px = tx * 8 + (lane & 7);
    py = ty * 8 + (lane >> 3);
    return px < P.W && py < P.H;
}

} // namespace ptx
