// ptx_hip.hip -- libptx_hip.so: C ABI (include/ptx.h) + kernels of the MI355X progressive path tracer.
//
// Build (see csrc/Makefile):  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
// -ffp-contract=off is part of the numerical contract (ptx_device.h): both the device code and the
// host-side precomputation below (edges, light pdf, aspect) must round every operation separately.
#include "../../include/ptx.h"
#include "ptx_wave.h"
#include "ptx_pool.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <limits>
#include <mutex>

using namespace ptx;

// ------------------------------------------------------------------------------------------ kernels

// Primary-hit cache: one thread per pixel, 8x8 tiles (glsl:501-514 + 207-249).
template <int PRUNE>
__global__ __launch_bounds__(64) void k_primary(DScene S, DParams P, float* __restrict__ prim_t, int* __restrict__ prim_tri,
                                                float4* __restrict__ dirs) {
    uint32_t px, py;
    if (!tile_pixel(P, blockIdx.x, threadIdx.x, px, py)) return;
    const V3 d = primary_dir(P, px, py);
    LocalCnt c = {};
    float t; int tri;
    closest_hit<PRUNE, false>(S, P.src, d, t, tri, c);
    const size_t i = (size_t)py * P.W + px;
    prim_t[i] = t;
    prim_tri[i] = tri;
    if (dirs) dirs[i] = make_float4(d.x, d.y, d.z, 0.f);
}

// Test hook: closest hit of arbitrary rays (glsl:207-249), one thread per ray.
template <int PRUNE, bool PAIRS>
__global__ __launch_bounds__(64) void k_trace(DScene S, const float* __restrict__ o, const float* __restrict__ d, uint32_t n,
                                              float* __restrict__ t_out, int* __restrict__ tri_out, unsigned long long* counters) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    LocalCnt c = {};
    if (i < n) {
        float t; int tri;
        c.rays = c.rays_traced = 1;
        const V3 ro = mk(o[3 * (size_t)i], o[3 * (size_t)i + 1], o[3 * (size_t)i + 2]), rd = mk(d[3 * (size_t)i], d[3 * (size_t)i + 1], d[3 * (size_t)i + 2]);
        if (PAIRS ? closest_hit_pairs<PRUNE, true>(S, ro, rd, t, tri, c) : closest_hit<PRUNE, true>(S, ro, rd, t, tri, c)) c.ray_hits = 1;
        t_out[i] = t;
        tri_out[i] = tri;
    }
    flush_counters(c, counters, 0);
}

// The path tracer: one thread per (pixel, frame-in-batch); one wave = one 8x8 pixel tile of one frame.
// Writes the frame's clamp(mean) into fb[frame][pixel]; k_accumulate folds the batch in frame order.
template <int PRUNE, bool MEMO, bool COUNT>
__global__ __launch_bounds__(64) void k_render(DScene S, DParams P, const int32_t* __restrict__ timestamps,
                                               const float* __restrict__ prim_t, const int* __restrict__ prim_tri,
                                               float4* __restrict__ fb, unsigned long long* counters) {
    uint32_t px, py;
    const bool valid = tile_pixel(P, blockIdx.x, threadIdx.x, px, py);
    const uint32_t frame = blockIdx.y;
    LocalCnt c = {};
    if (valid) {
        const size_t i = (size_t)py * P.W + px;
        const V3 rgb = render_pixel_frame<PRUNE, MEMO, COUNT>(S, P, px, py, timestamps[frame], prim_t[i], prim_tri[i], c);
        fb[(size_t)frame * P.W * P.H + i] = make_float4(rgb.x, rgb.y, rgb.z, 1.0f);
    }
    flush_counters(c, counters, valid ? 1u : 0u);
}

// accum[p] = ((accum[p] + fb[0][p]) + fb[1][p]) + ...   (frame order => independent of the batch size)
__global__ __launch_bounds__(256) void k_accumulate(float4* __restrict__ accum, const float4* __restrict__ fb, uint32_t npix, uint32_t nframes) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= npix) return;
    float4 a = accum[p];
    for (uint32_t f = 0; f < nframes; f++) {
        const float4 v = fb[(size_t)f * npix + p];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    accum[p] = a;
}

// Display pass (shaders/fragmentShaderOutput.glsl:22-41) + RGBA8 pack + optional flip (main.js:464-473).
__global__ __launch_bounds__(256) void k_pack_rgba8(const float4* __restrict__ accum, uint32_t W, uint32_t H, float inv_frames_is_div,
                                                    int apply_gamma, int flip_y, uchar4* __restrict__ out) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= W * H) return;
    const uint32_t x = p % W, y = p / W;
    const float4 a = accum[p];
    float v[3] = {a.x / inv_frames_is_div, a.y / inv_frames_is_div, a.z / inv_frames_is_div};
    unsigned char q[3];
    for (int k = 0; k < 3; k++) {
        float cval = clamp01(v[k]);
        if (apply_gamma) cval = ss_pow(cval, 1.0f / 2.2f);
        q[k] = (unsigned char)floorf(clamp01(cval) * 255.0f + 0.5f);
    }
    const uint32_t oy = flip_y ? (H - 1 - y) : y;
    out[(size_t)oy * W + x] = make_uchar4(q[0], q[1], q[2], 255);
}

// Reference-compatible progressive blend (option "compat_rgba8"): glsl:532-552 through the RGBA8 ping-pong render targets
// of modules/bufferManager.js:18 -- out = unorm8(mix(prev, cur, 1 / (frameNumber + 1))), where prev is read back from
// the 8-bit target with the shader's aspect-shifted texture coordinate (glsl:501-512; NEAREST, REPEAT wrap).
// rules: what GL leaves to the implementation.  1 = the preferred ones (store round(c * 255); texel floor(coord * size));
// 2 = SwiftShader's, the implementation the reference's committed frames come from (store through a truncated 16-bit
// intermediate; texel (trunc(coord * 65536) * size) >> 16), under which those frames are reproduced byte for byte.
__global__ __launch_bounds__(256) void k_blend_rgba8(const uchar4* __restrict__ prev, const float4* __restrict__ cur, uchar4* __restrict__ out,
                                                     uint32_t W, uint32_t H, float aspect, float bf, int rules) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= W * H) return;
    const uint32_t x = p % W, y = p / W;
    float u = ((float)x + 0.5f) / (float)W, v = ((float)y + 0.5f) / (float)H;
    if (aspect > 1.0f) { u = u * aspect; u = u - 0.125f; }
    else if (aspect < 1.0f) { v = v / aspect; v = v - 0.125f; }
    const float fu = u - floorf(u), fv = v - floorf(v);
    int sx = (int)floorf(fu * (float)W), sy = (int)floorf(fv * (float)H);
    if (rules == 2) {
        const uint32_t u16 = min((uint32_t)(fu * 65536.0f), 65535u), v16 = min((uint32_t)(fv * 65536.0f), 65535u);
        sx = (int)(((uint64_t)u16 * W) >> 16); sy = (int)(((uint64_t)v16 * H) >> 16);
    }
    if (sx >= (int)W) sx = (int)W - 1;
    if (sy >= (int)H) sy = (int)H - 1;
    const uchar4 q = prev[(size_t)sy * W + sx];
    const float4 c = cur[p];
    const float pr[3] = {clamp01((float)q.x / 255.0f), clamp01((float)q.y / 255.0f), clamp01((float)q.z / 255.0f)};
    const float cu[3] = {c.x, c.y, c.z};
    unsigned char o[3];
    for (int k = 0; k < 3; k++) {
        const float b = pr[k] * (1.0f - bf) + cu[k] * bf;      // mix(x, y, a) = x * (1 - a) + y * a
        if (rules == 2) {
            const uint32_t c16 = (uint32_t)(clamp01(b) * 65535.0f);
            o[k] = (unsigned char)((c16 - (c16 >> 8) + 0x80u) >> 8);
        } else {
            o[k] = (unsigned char)floorf(clamp01(b) * 255.0f + 0.5f);
        }
    }
    out[p] = make_uchar4(o[0], o[1], o[2], 255);
}

// display pass over the 8-bit target (shaders/fragmentShaderOutput.glsl:35) + optional flip (main.js:464-473)
// rules: the canvas' float -> 8-bit store, as in k_blend_rgba8 (1 = round to nearest, 2 = SwiftShader's 16-bit intermediate)
__global__ __launch_bounds__(256) void k_display_rgba8(const uchar4* __restrict__ in, uint32_t W, uint32_t H, int apply_gamma, int flip_y, int rules,
                                                       uchar4* __restrict__ out) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    if (p >= W * H) return;
    const uint32_t x = p % W, y = p / W;
    const uchar4 q = in[p];
    unsigned char o[3] = {q.x, q.y, q.z};
    if (apply_gamma)
        for (int k = 0; k < 3; k++) {
            const float g = clamp01(ss_pow((float)o[k] / 255.0f, 1.0f / 2.2f));
            if (rules == 2) {
                const uint32_t c16 = (uint32_t)(g * 65535.0f);
                o[k] = (unsigned char)((c16 - (c16 >> 8) + 0x80u) >> 8);
            } else {
                o[k] = (unsigned char)floorf(g * 255.0f + 0.5f);
            }
        }
    const uint32_t oy = flip_y ? (H - 1 - y) : y;
    out[(size_t)oy * W + x] = make_uchar4(o[0], o[1], o[2], 255);
}

// ---- static scratch of ptx_render_frames_noalloc: the entry point that allocates nothing (include/ptx.h)
// One array per GPU the library runs on, held until the process ends: a 512-B header (task counter of pipeline 2 at byte 0,
// the timestamps of one batch at byte 256) followed by the frame buffers of one batch.  In host code the name g_scratch is
// only a stand-in for the symbol: the device address is taken per device with hipGetSymbolAddress (scratch_for_device) and
// every use below goes through that pointer.
#define PTX_SCRATCH_HEADER 512u
#define PTX_SCRATCH_TS_OFFSET 256u
#define PTX_SCRATCH_MAX_BATCH 64u                       // timestamps travel by value in one kernel argument block
#define PTX_SCRATCH_FB_BYTES (64u << 20)                // 64 MiB = 4 194 304 (pixel, frame) results of 16 B
static __device__ __align__(256) unsigned char g_scratch[PTX_SCRATCH_HEADER + PTX_SCRATCH_FB_BYTES];

struct TsBlock { int32_t v[PTX_SCRATCH_MAX_BATCH]; };
// timestamps of a batch, passed by value, into the scratch (no host-to-device copy, nothing to stage)
__global__ __launch_bounds__(64) void k_store_timestamps(TsBlock ts, int32_t* __restrict__ dst, uint32_t n) {
    if (threadIdx.x < n) dst[threadIdx.x] = ts.v[threadIdx.x];
}

// ---- ptx_selfcheck_pruning: the certificate behind option "traversal" bit 0 (DESIGN.md "Pruning and exactness")
// Every thread generates one ADVERSARIAL ray from a counter-based hash (no host traffic, so 10^8 rays take a fraction of a
// second) and traces it twice through the same tree: once without pruning (exact by construction) and once with it.  Any
// difference in (t bits, triangle) is counted and the first 16 are recorded.  Families (each biased towards what could
// break a best-t cut: hits whose Moeller-Trumbore t is ill-conditioned, boxes entered at grazing angles, ties):
//   0 surface point -> uniform direction            4 skimming a triangle's bounding-box face from just outside the slab
//   1 surface point -> point on another triangle    5 grazing directions of the hemisphere about the surface's own normal
//   2 in the plane of the own triangle +- tiny tilt 6 (nearly) axis-parallel rays: huge / infinite 1/d components
//   3 grazing a LARGE triangle from far away        7 surface point -> vertex / edge of another triangle
//   8 .. 11 = families 1, 3, 4, 7 aimed at the EMISSIVE triangles only: what the any-hit rule for light-visibility rays rests
//   on.  For every ray (of any family) whose unpruned closest hit is an emissive triangle at t0 the kernel also checks that
//   light_bound does not claim "no emissive triangle can be hit at or before t0" -- the only way that rule could end a ray
//   on a blocker although an emitter is nearer (a blocker's t_b >= t0, and the claim is monotone in t).
struct CheckArgs { unsigned long long n; uint32_t seed, family, n_big, n_emit; const uint32_t* big; const uint32_t* emit; V3 bmin, bmax; };
struct CheckOut { unsigned long long rays, hits, mismatches; float first[16][8]; int first_tri[16][2]; unsigned long long emitter_hits, anyhit_violations; };

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
struct CheckRng {
    uint32_t s;
    __device__ __forceinline__ uint32_t u() { s = hash32(s + 0x9E3779B9u); return s; }
    __device__ __forceinline__ float f() { return (float)(u() >> 8) * (1.0f / 16777216.0f); }              // [0, 1)
    __device__ __forceinline__ float sgn() { return (u() & 1u) ? 1.0f : -1.0f; }
    __device__ __forceinline__ float tiny(float lo_exp, float span) { return exp2f(-(lo_exp + span * f())); }  // log-uniform 2^-lo .. 2^-(lo+span)
};

template <int PRUNE, bool PAIRS>
__global__ __launch_bounds__(64) void k_selfcheck(DScene S, CheckArgs A, CheckOut* out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 64ull + threadIdx.x;
    if (i >= A.n) return;
    CheckRng R{hash32(A.seed ^ hash32((uint32_t)i) ^ hash32((uint32_t)(i >> 32) + 0x51ED27u) ^ (A.family * 0x9E3779B1u))};
    const uint32_t T = (uint32_t)S.T;
    auto tri = [&](uint32_t k, V3& v0, V3& e1, V3& e2) { v0 = mk(ld16(S.tris, k * 48u)); e1 = mk(ld16(S.tris, k * 48u + 16u)); e2 = mk(ld16(S.tris, k * 48u + 32u)); };
    auto point_on = [&](V3 v0, V3 e1, V3 e2) { float a = R.f(), b = R.f(); if (a + b > 1.0f) { a = 1.0f - a; b = 1.0f - b; } return (v0 + e1 * a) + e2 * b; };
    const bool at_emitters = A.family >= 8u && A.n_emit != 0u;
    const uint32_t family = A.family >= 8u ? (A.family == 8u ? 1u : (A.family == 9u ? 3u : (A.family == 10u ? 4u : 7u))) : A.family;
    auto pick = [&](bool prefer_big) { return at_emitters ? A.emit[R.u() % A.n_emit] : ((prefer_big && A.n_big && (R.u() & 1u)) ? A.big[R.u() % A.n_big] : R.u() % T); };
    const V3 ext = A.bmax - A.bmin;
    const float scale = fmaxf(fmaxf(ext.x, ext.y), ext.z);
    V3 a0, a1, a2;
    tri(R.u() % T, a0, a1, a2);
    V3 o = point_on(a0, a1, a2), d = mk(1.f, 0.f, 0.f);
    const V3 nA = normalize(cross(a1, a2));
    auto in_plane = [&](V3 e1, V3 e2) { return normalize(e1 * (R.f() - 0.5f) + e2 * (R.f() - 0.5f)); };
    switch (family) {
        case 0: {
            const float z = 2.0f * R.f() - 1.0f, ph = 6.2831853f * R.f(), r = sqrtf(fmaxf(0.f, 1.0f - z * z));
            d = mk(r * cosf(ph), r * sinf(ph), z);
            break;
        }
        case 1: { V3 b0, b1, b2; tri(pick(true), b0, b1, b2); d = normalize(point_on(b0, b1, b2) - o); break; }
        case 2: d = normalize(in_plane(a1, a2) + nA * (R.sgn() * R.tiny(1.f, 22.f))); break;
        case 3: {
            V3 b0, b1, b2; tri(pick(true), b0, b1, b2);
            const V3 tgt = point_on(b0, b1, b2), w = in_plane(b1, b2), nB = normalize(cross(b1, b2));
            const float L = scale * exp2f(-6.0f * R.f()), sn = R.tiny(4.f, 20.f);
            o = (tgt - w * L) + nB * (R.sgn() * L * sn);
            d = normalize(tgt - o);
            break;
        }
        case 4: {
            V3 b0, b1, b2; tri(pick(false), b0, b1, b2);
            const V3 p1 = b0 + b1, p2 = b0 + b2;
            const V3 lo = mk(fminf(fminf(b0.x, p1.x), p2.x), fminf(fminf(b0.y, p1.y), p2.y), fminf(fminf(b0.z, p1.z), p2.z));
            const V3 hi = mk(fmaxf(fmaxf(b0.x, p1.x), p2.x), fmaxf(fmaxf(b0.y, p1.y), p2.y), fmaxf(fmaxf(b0.z, p1.z), p2.z));
            const V3 tgt = point_on(b0, b1, b2);
            const uint32_t k = R.u() % 3u;
            const bool upper = (R.u() & 1u) != 0;
            const float delta = scale * R.tiny(6.f, 18.f) * (upper ? 1.0f : -1.0f), L = scale * exp2f(-8.0f * R.f());
            const float ph = 6.2831853f * R.f(), c = L * cosf(ph), sn = L * sinf(ph);
            o = tgt;                                   // lateral offset in the two other axes, a hair outside the slab of axis k
            if (k == 0) { o.x = (upper ? hi.x : lo.x) + delta; o.y += c; o.z += sn; }
            else if (k == 1) { o.y = (upper ? hi.y : lo.y) + delta; o.z += c; o.x += sn; }
            else { o.z = (upper ? hi.z : lo.z) + delta; o.x += c; o.y += sn; }
            d = normalize(tgt - o);
            break;
        }
        case 5: { const float z = R.tiny(1.f, 22.f); d = normalize(in_plane(a1, a2) * sqrtf(1.0f - z * z) + nA * z); break; }
        case 6: {
            const uint32_t k = R.u() % 3u, mode = R.u() % 3u;
            const float e1 = mode == 0 ? 0.f : R.sgn() * R.tiny(10.f, 110.f), e2 = mode == 2 ? R.sgn() * R.tiny(10.f, 110.f) : 0.f, m = R.sgn();
            d = k == 0 ? mk(m, e1, e2) : (k == 1 ? mk(e2, m, e1) : mk(e1, e2, m));
            if (R.u() & 1u) o = mk(A.bmin.x + ext.x * R.f(), A.bmin.y + ext.y * R.f(), A.bmin.z + ext.z * R.f());
            break;
        }
        default: {
            V3 b0, b1, b2; tri(pick(true), b0, b1, b2);
            const uint32_t w = R.u() % 6u;
            const float s = R.f();
            V3 tgt = w == 0 ? b0 : (w == 1 ? b0 + b1 : (w == 2 ? b0 + b2 : (w == 3 ? b0 + b1 * s : (w == 4 ? b0 + b2 * s : (b0 + b1 * s) + b2 * (1.0f - s)))));
            d = normalize(tgt - o);
            break;
        }
    }
    LocalCnt c = {};
    float t0, t1;
    int k0, k1;
    bool h0;
    if (PAIRS) { h0 = closest_hit_pairs<0, false>(S, o, d, t0, k0, c); closest_hit_pairs<PRUNE, false>(S, o, d, t1, k1, c); }
    else { h0 = closest_hit<0, false>(S, o, d, t0, k0, c); closest_hit<PRUNE, false>(S, o, d, t1, k1, c); }
    const bool bad = __float_as_uint(t0) != __float_as_uint(t1) || k0 != k1;
    // the any-hit rule: an emitter that IS the closest hit must never be excluded by light_bound at its own distance
    bool emitter = false, violation = false;
    if (h0 && S.n_lbox != 0u) {
        const float4 em = S.attrE[k0];
        emitter = em.x != 0.0f || em.y != 0.0f || em.z != 0.0f;
        const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        if (emitter && !inv_degenerate(inv)) {
            Kappa kap = {0.f, 0.f};
            if (PRUNE == 1) kap = unpack_kappa(pack_kappa(ray_kappa(S, o, d)));
            violation = light_settles<PRUNE>(S, light_bound<PRUNE>(S, o, d, inv, kap), t0, kap);
        }
    }
    const unsigned long long mh = __ballot(h0), mb = __ballot(bad), ma = __ballot(1), me = __ballot(emitter), mv = __ballot(violation);
    if (first_active_lane()) {
        atomicAdd(&out->rays, (unsigned long long)__popcll(ma));
        atomicAdd(&out->hits, (unsigned long long)__popcll(mh));
        if (mb) atomicAdd(&out->mismatches, (unsigned long long)__popcll(mb));
        if (me) atomicAdd(&out->emitter_hits, (unsigned long long)__popcll(me));
        if (mv) atomicAdd(&out->anyhit_violations, (unsigned long long)__popcll(mv));
    }
    if (bad) {
        for (int q = 0; q < 16; q++) {                       // claim the first free record
            if (atomicCAS(&out->first_tri[q][0], -2, k0) == -2) {
                out->first_tri[q][1] = k1;
                const float rec[8] = {o.x, o.y, o.z, d.x, d.y, d.z, t0, t1};
                for (int e = 0; e < 8; e++) out->first[q][e] = rec[e];
                break;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ host side

// Two launch slots, used alternately by successive ptx_render_frames calls: each has its own stream, frame buffers,
// timestamps and task counter, so the path-tracing kernel of call k + 1 starts filling the chip while the last, longest paths
// of call k are still finishing (one-frame-per-call loops like main.js:400-405 otherwise pay that drain on every frame).
// The frame-order sum into the shared accumulation buffer is kept by events: a call's k_accumulate waits for the other slot's
// last one.
struct LaunchSlot {
    hipStream_t stream = nullptr;
    float4* d_fb = nullptr;                     // batch frame buffers
    size_t fb_frames = 0;
    int32_t* d_ts = nullptr;
    size_t ts_cap = 0;
    uint32_t* d_task_counter = nullptr;         // pipeline 2: next undrawn (pixel, frame) task of the running launch
    hipEvent_t ev_done = nullptr;               // recorded behind the slot's last accumulate / blend
    bool busy = false;                          // ev_done has been recorded since the last full synchronisation
};

struct ptx_ctx {
    int device = -1;
    hipStream_t stream = nullptr;               // = slot[0].stream: everything that is not a render launch runs here
    LaunchSlot slot[2];
    int next_slot = 0;
    int opt_anyhit = 1;                         // option "anyhit": light-visibility rays stop at the first conclusive blocker (pipeline 2, pruning on)
    int opt_overlap = 1;                        // option "overlap": 0 = every call on slot 0 (strictly one launch after the other)
    std::string err;

    // scene
    bool have_scene = false;
    DScene S = {};
    const float4* nodes_sah = nullptr;   // leaf-SAH tree over the reference leaves: 8 direction-octant layouts + the reference tree, or nullptr
    int N_sah = 0;
    std::vector<void*> scene_allocs;
    float bbox_min[3] = {0, 0, 0}, bbox_max[3] = {0, 0, 0};   // of all triangles (ptx_selfcheck_pruning aims rays with it)
    const uint32_t* d_big_slots = nullptr;                     // inorder slots of the largest triangles, by area
    const uint32_t* d_emit_slots = nullptr;                    // inorder slots of the triangles with a non-zero emission (selfcheck families 8 .. 11)
    uint32_t n_emit_slots = 0;
    const float4* d_emit_list = nullptr;                       // provable any-hit rule: (e2 x e1) / n per emissive triangle (at most PTX_MAX_EMIT; ptx_kernels.h light_bound)
    uint32_t n_emit_list = 0;
    float emit_c1 = 0.f;
    int opt_margin_lat = PTX_MARGIN_LAT_PPM;                   // option "margin_lat": the margin rule's lateral allowance in ppm of the scene's diagonal (0 = round 2's rule)
    int opt_emit_list = 1;                                     // option "emit_list": 0 = the provable rule bounds the emitters by their largest coefficients only (round 2)
    uint32_t n_big_slots = 0;

    // params
    bool have_params = false;
    DParams P = {};
    bool prim_valid = false;
    float* d_prim_t = nullptr;
    int* d_prim_tri = nullptr;
    float4* d_accum = nullptr;
    uchar4* d_compat[2] = {nullptr, nullptr};   // option "compat_rgba8": the reference's two RGBA8 render targets
    int compat_cur = 0;                         // which of the two holds the newest blend
    uint32_t frames_done = 0;

    unsigned char* d_scratch = nullptr;         // this device's copy of g_scratch (ptx_render_frames_noalloc); not owned
    void* d_stage = nullptr;                    // grow-only staging for read-backs and the trace hook (ensure_stage)
    size_t stage_cap = 0;

    // options
    int opt_prune = 2;   // 2 = heuristic margin (default), 1 = provable pruning (option "prune"); applies when traversal bit 0 is set
    int opt_traversal = 3, opt_memo = 1, opt_count = 0, opt_pipeline = 2, opt_frame_batch = 0, opt_tasks_per_wave = 128, opt_compat = 0, opt_pool_chunk = 64;

    // multi-GPU (DESIGN.md "Multi-GPU")
    std::vector<ptx_ctx*> subs;                 // ptx_create with n_devices > 1: one single-device context per device id; this object only routes
    bool same_device = false;                   // all subs on ONE device (rehearsal on a 1-GPU box): summed by k_sum_accums, no RCCL
    std::vector<void*> group_comms;             // ncclComm_t per sub (ncclCommInitAll), created at the first reduction
    bool total_valid = false;                   // subs[0]->d_total holds the sum of the subs' accumulations as they are now
    float4* d_total = nullptr;                  // W*H float4: destination of the all-reduce (sub contexts; ptx_allreduce with d_dst == NULL stays in place)
    void* comm = nullptr;                       // one process per GPU: ncclComm_t of ptx_comm_init
    int comm_world = 0, comm_rank = -1;
    unsigned long long* d_comm_cnt = nullptr;   // frames_done travels through the same communicator

    // counters + timing
    unsigned long long* d_counters = nullptr;
    uint64_t pool_resident[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // pipeline 2: grid size per kernel variant (queried once)
    bool prune_ok = true;                      // pruning, any-hit and the blocker cache need nested boxes that contain their triangles
    bool pool_ok = true;                       // pipeline 2 needs disjoint leaf ranges (pair records are addressed by first slot)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events;
    std::vector<hipEvent_t> free_events;
    double kernel_ms = 0.0;
    uint64_t kernel_launches = 0;
};

static std::string g_create_error;

// Device address of g_scratch on each GPU (every device holds its own copy of the array at its own address), taken once
// per device in ptx_create with that device current.
#define PTX_MAX_DEVICES 64
static std::mutex g_scratch_mutex;
static unsigned char* g_scratch_ptr[PTX_MAX_DEVICES] = {};
static hipError_t scratch_for_device(int device, unsigned char** out) {   // `device` must be the current device
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (device < 0 || device >= PTX_MAX_DEVICES) return hipErrorInvalidDevice;
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_scratch));
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}

// traversal option: bit 0 = best-t pruning, bit 1 = walk the leaf-SAH tree instead of the reference tree
static inline int prune_mode(const ptx_ctx* c) { return ((c->opt_traversal & 1) && c->prune_ok) ? c->opt_prune : 0; }   // the kernels' PRUNE parameter
static inline float __uint_as_float_host(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline DScene scene_view(const ptx_ctx* c) {
    DScene S = c->S;
    if ((c->opt_traversal & 2) && c->nodes_sah) { S.nodes = c->nodes_sah; S.N = c->N_sah; S.oct_stride = (uint32_t)c->N_sah * 32u; S.ref_base = 8u * S.oct_stride; }
    S.lat = (float)c->opt_margin_lat * 1.0e-6f * (2.0f * c->S.rs);   // margin rule: lateral allowance per unit of max|1/d_k| (ptx_kernels.h)
    if (!c->opt_anyhit) S.n_lbox = 0;              // (kernels without pruning never look at the boxes)
    // provable rule: with emitters whose position-error coefficient is this large the bound never lets a ray stop (pixar_room's
    // lamp: c_o = 0.054, slack = c_o * kappa of several scene units), and computing it would only cost time
    // -- unless the scene has few enough emitters for the per-ray bound (light_bound, emitter list), which does not depend on it
    S.emit = nullptr; S.n_emit = 0; S.emit_c1 = 0.f;
    if (prune_mode(c) == 1) {
        if (c->d_emit_list && c->n_emit_list > 0 && c->opt_emit_list) { S.emit = c->d_emit_list; S.n_emit = c->n_emit_list; S.emit_c1 = c->emit_c1; }
        else if (!(__uint_as_float_host(S.lcoefs & 0x7FFF0000u) <= 0.01f)) S.n_lbox = 0;
    }
    return S;
}

#define HIP_TRY(ctx, expr)                                                                               \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                              \
            return PTX_ERR_HIP;                                                                          \
        }                                                                                                \
    } while (0)

static int fail(ptx_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// The static scratch of a GPU has ONE user at a time: the context whose ptx_render_frames_noalloc work may still be running
// on it.  Claimed by that entry point, released when the context has synchronised (sync_all) or is destroyed; a second
// context that asks meanwhile gets PTX_ERR_STATE instead of a race.
static const ptx_ctx* g_scratch_owner[PTX_MAX_DEVICES] = {};
static bool scratch_claim(const ptx_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (ctx->device < 0 || ctx->device >= PTX_MAX_DEVICES) return false;
    if (g_scratch_owner[ctx->device] && g_scratch_owner[ctx->device] != ctx) return false;
    g_scratch_owner[ctx->device] = ctx;
    return true;
}
static void scratch_release(const ptx_ctx* ctx) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (ctx->device >= 0 && ctx->device < PTX_MAX_DEVICES && g_scratch_owner[ctx->device] == ctx) g_scratch_owner[ctx->device] = nullptr;
}

// both launch slots idle (device current)
static hipError_t sync_all(ptx_ctx* ctx) {
    hipError_t e = hipSuccess;
    for (int k = 0; k < 2; k++)
        if (ctx->slot[k].stream) { const hipError_t ek = hipStreamSynchronize(ctx->slot[k].stream); if (e == hipSuccess) e = ek; ctx->slot[k].busy = false; }
    if (e == hipSuccess) scratch_release(ctx);
    return e;
}

// ---- multi-device contexts route every entry point to their subs (definitions at the end of this file)
static int multi_upload_scene(ptx_ctx*, const ptx_scene*);
static int multi_set_params(ptx_ctx*, const ptx_params*);
static int multi_set_option(ptx_ctx*, const char*, int64_t);
static int multi_render_frames(ptx_ctx*, const int32_t*, uint32_t, uint32_t, bool noalloc);
static int multi_for_all(ptx_ctx*, int (*fn)(ptx_ctx*));
static int multi_frames_done(const ptx_ctx*);
static int multi_read(ptx_ctx*, int kind, void* out, int a, int b);
static int multi_write_sum(ptx_ctx*, const float*, uint32_t);
static int multi_get_counters(ptx_ctx*, ptx_counters*);
static int multi_accum_export(ptx_ctx*, void*);
static int multi_accum_import(ptx_ctx*, const void*, uint32_t);
static void multi_destroy(ptx_ctx*);
static ptx_ctx* multi_create(const int* device_ids, int n_devices);
static void comm_release(ptx_ctx*);
#define PTX_MULTI(ctx, call) do { if ((ctx) && !(ctx)->subs.empty()) return (call); } while (0)
enum { RD_LINEAR = 0, RD_SUM, RD_RGBA8, RD_COMPAT };

extern "C" int ptx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char* ptx_last_error(const ptx_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" void ptx_destroy(ptx_ctx* ctx);

extern "C" ptx_ctx* ptx_create(const int* device_ids, int n_devices) {
    if (!device_ids || n_devices < 1 || n_devices > 64) {
        g_create_error = "ptx_create: device_ids must list 1 .. 64 devices";
        return nullptr;
    }
    if (n_devices > 1) return multi_create(device_ids, n_devices);
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("ptx_create: no HIP device: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (device_ids[0] < 0 || device_ids[0] >= count) {
        g_create_error = "ptx_create: device id out of range";
        return nullptr;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_ids[0])) != hipSuccess) {
        g_create_error = std::string("ptx_create: hipGetDeviceProperties: ") + hipGetErrorString(e);
        return nullptr;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("ptx_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return nullptr;
    }
    ptx_ctx* ctx = new ptx_ctx();
    ctx->device = device_ids[0];
    if ((e = hipSetDevice(ctx->device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->slot[0].stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->slot[1].stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->slot[0].ev_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->slot[1].ev_done, hipEventDisableTiming)) != hipSuccess ||
        (e = hipMalloc(&ctx->d_counters, CNT_NUM * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMemset(ctx->d_counters, 0, CNT_NUM * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc(&ctx->slot[0].d_task_counter, sizeof(uint32_t))) != hipSuccess ||
        (e = hipMalloc(&ctx->slot[1].d_task_counter, sizeof(uint32_t))) != hipSuccess ||
        (e = scratch_for_device(ctx->device, &ctx->d_scratch)) != hipSuccess) {
        g_create_error = std::string("ptx_create: ") + hipGetErrorString(e);
        ctx->stream = ctx->slot[0].stream;
        ptx_destroy(ctx);                  // releases whatever was created before the failing call
        return nullptr;
    }
    ctx->stream = ctx->slot[0].stream;
    return ctx;
}

static void free_scene(ptx_ctx* ctx) {
    for (void* p : ctx->scene_allocs) (void)hipFree(p);
    ctx->scene_allocs.clear();
    ctx->have_scene = false;
    ctx->prim_valid = false;
}

static void free_params_buffers(ptx_ctx* ctx) {
    if (ctx->d_prim_t) (void)hipFree(ctx->d_prim_t);
    if (ctx->d_prim_tri) (void)hipFree(ctx->d_prim_tri);
    if (ctx->d_accum) (void)hipFree(ctx->d_accum);
    for (int k = 0; k < 2; k++) { if (ctx->d_compat[k]) (void)hipFree(ctx->d_compat[k]); ctx->d_compat[k] = nullptr; }
    for (LaunchSlot& sl : ctx->slot) { if (sl.d_fb) (void)hipFree(sl.d_fb); sl.d_fb = nullptr; sl.fb_frames = 0; }
    if (ctx->d_total) (void)hipFree(ctx->d_total);
    ctx->d_total = nullptr;
    ctx->d_prim_t = nullptr; ctx->d_prim_tri = nullptr; ctx->d_accum = nullptr;
    ctx->prim_valid = false;
    ctx->have_params = false;                  // set again only once every buffer of the new size exists
}

// Grow-only device staging shared by the read-backs and ptx_trace_closest (one buffer, reused: no hipMalloc per call).
static int ensure_stage(ptx_ctx* ctx, size_t bytes) {
    if (ctx->stage_cap >= bytes) return PTX_OK;
    HIP_TRY(ctx, sync_all(ctx));
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    ctx->d_stage = nullptr; ctx->stage_cap = 0;
    HIP_TRY(ctx, hipMalloc(&ctx->d_stage, bytes));
    ctx->stage_cap = bytes;
    return PTX_OK;
}

extern "C" void ptx_destroy(ptx_ctx* ctx) {
    if (!ctx) return;
    if (!ctx->subs.empty()) { multi_destroy(ctx); return; }
    (void)hipSetDevice(ctx->device);
    (void)sync_all(ctx);
    scratch_release(ctx);
    comm_release(ctx);
    free_scene(ctx);
    free_params_buffers(ctx);
    if (ctx->d_stage) (void)hipFree(ctx->d_stage);
    if (ctx->d_counters) (void)hipFree(ctx->d_counters);
    for (auto& pr : ctx->pending_events) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto ev : ctx->free_events) (void)hipEventDestroy(ev);
    for (LaunchSlot& sl : ctx->slot) {
        if (sl.d_ts) (void)hipFree(sl.d_ts);
        if (sl.d_task_counter) (void)hipFree(sl.d_task_counter);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
    }
    delete ctx;
}

template <typename T>
static int upload(ptx_ctx* ctx, const std::vector<T>& h, const T** out) {
    void* p = nullptr;
    const size_t bytes = (h.empty() ? 1 : h.size()) * sizeof(T);
    HIP_TRY(ctx, hipMalloc(&p, bytes));
    ctx->scene_allocs.push_back(p);
    if (!h.empty()) HIP_TRY(ctx, hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return PTX_OK;
}

// ---------------------------------------------------------------------------------- leaf-SAH tree (traversal bit 1)
// The reference's closest hit only depends on WHICH LEAVES have their own box hit: every ancestor box contains its
// children's boxes exactly (unions taken in f64, each rounded once to f32; rounding is monotone), and the slab test
// (glsl:175-191) is monotone in the box bounds operation by operation (sub, mul, min, max are all monotone, and with
// minNum/maxNum a 0*inf = NaN slab can only occur for the smaller box), so "leaf box hit" implies "all ancestors hit".
// The hierarchy above the leaves is therefore free: we rebuild it with a surface-area heuristic over the reference
// leaves' exact boxes, taking exact float min/max unions for the new inner boxes, and the same threaded walk finds
// exactly the same set of leaves with about half the box tests on pixar_room.  Leaves are visited in a different order,
// so the winner among equal t is chosen by inorder slot explicitly (closest_hit's (t, k) compare).
// Returns false when the reference arrays do not have that containment property (then the reference tree is walked).

// ---- provable pruning (ptx_kernels.h, PRUNE = 1): the position-error coefficients of a triangle.
// Claim (proof in DESIGN.md "Pruning and exactness"; u = 2^-24, strict binary32, the operation order of tri_test): let the
// ray satisfy | |d|^2 - 1 | <= 1e-3 and let tri_test / tri_test2 (glsl:150-173) ACCEPT a hit on the triangle (v0, e1, e2)
// with computed determinant a^ and distance t^.  With E = max(|e1|, |e2|) and any M >= max(|o - v0|, E):
//     if 61 u M E <= |a^|:   the point o + t^ d lies within   (u / |a^|) * 90 E^2 M + 20 u M   of the triangle.
// (Absolute errors: a^ within 7u E^2 of d . (e2 x e1); the three numerators within 8.1u M E^2 / {1, |e1|, |e2|}; two more
// roundings in f = 1 / a^ and the product; accepted barycentrics u^, v^ >= 0, u^ + v^ <= 1 + 2u.)  Two ways to bound |a^|:
//   * any accepted hit has |a^| >= eps0 = 0.0001f (glsl:156)             -> c_o = 1.01 (90 (u / eps0) E^2 + 20 u),
//     valid when 61 (u / eps0) M_cap E <= 1, else inf (never pruned);
//   * a triangle whose unit normal is within delta of +- N_j (a "class") has |a^| >= 2 area (|d . N_j| - delta') >= 2 area / w_ray
//                                                                        -> c_c = 1.01 (90 u E^2 / (2 area) + 20 u), times w_ray,
//     valid when E^2 / (2 area) <= 10 and 61 u M_cap E WCAP / (2 area) <= 1.
// Every triangle is charged to ONE of the two: c_c when it is a class member and that is the smaller slack at w_ray = 1.
struct TriCoef { float c_o, c_c; };
static const double kU = 5.9604644775390625e-08, kEps0 = (double)0.0001f;
static float coef_other_of(double E, double m_cap) {
    if (!(61.0 * (kU / kEps0) * m_cap * E <= 1.0)) return INFINITY;
    return (float)(1.01 * (90.0 * (kU / kEps0) * E * E + 20.0 * kU));
}
static float coef_class_of(double E, double area2, double m_cap) {      // inf: cannot be a class member
    if (!(area2 > 0.0) || !(E * E / area2 <= 10.0) || !(61.0 * kU * m_cap * E * (double)PTX_CLS_WCAP / area2 <= 1.0)) return INFINITY;
    return (float)(1.01 * (90.0 * kU * E * E / area2 + 20.0 * kU));
}
// 16-bit float (sign 0, 8-bit exponent, 7-bit mantissa), rounded UP; two of them make the packed word the kernels decode
static uint32_t half_up(float c) {
    if (!(c >= 0.f)) return 0x7FFFu;                                     // NaN: never pruned
    uint32_t b; std::memcpy(&b, &c, 4);
    return std::min<uint32_t>((b >> 16) + ((b & 0xFFFFu) ? 1u : 0u), 0x7F80u);   // at most +inf
}
static uint32_t pack_coefs(TriCoef c) { return (half_up(c.c_o) << 16) | half_up(c.c_c); }    // bit 31 clear; inner nodes set it
struct SahItem { float mn[3], mx[3]; int fc; float cen[3]; double w; TriCoef coef; };

// max of v over [lo, hi) for many ranges: a flat segment tree (the nodes a threaded walk skips when node i is cut are
// exactly (i, miss[i]), whatever shape the caller's arrays have)
struct RangeMax {
    size_t n;
    std::vector<float> t;
    explicit RangeMax(const std::vector<float>& v) : n(v.size()), t(2 * v.size(), 0.f) {
        std::copy(v.begin(), v.end(), t.begin() + n);
        for (size_t i = n; i-- > 1;) t[i] = std::max(t[2 * i], t[2 * i + 1]);
    }
    float query(size_t lo, size_t hi) const {
        float m = 0.f;
        for (lo += n, hi += n; lo < hi; lo >>= 1, hi >>= 1) {
            if (lo & 1) m = std::max(m, t[lo++]);
            if (hi & 1) m = std::max(m, t[--hi]);
        }
        return m;
    }
};

static bool ref_tree_is_monotone(const ptx_scene* sc) {
    const uint32_t N = sc->N;
    std::vector<uint32_t> end(N, 0);
    // children of an inner node i: L = i + 1, R = miss[L]; the subtree of i ends at miss[i] (or N)
    for (uint32_t i = 0; i < N; i++) {
        const int32_t m = sc->node_miss[i];
        end[i] = m < 0 ? N : (uint32_t)m;
    }
    for (uint32_t i = 0; i < N; i++) {
        if (sc->node_first[i] != -2) { if (end[i] != i + 1 && !(sc->node_miss[i] == -1 && i + 1 == N)) return false; continue; }
        const uint32_t L = i + 1;
        if (L >= N) return false;
        const uint32_t R = end[L];
        if (R >= N || R <= L || end[R] != end[i]) return false;
        for (uint32_t c : {L, R})
            for (int k = 0; k < 3; k++)
                if (!(sc->node_min[3 * (size_t)i + k] <= sc->node_min[3 * (size_t)c + k]) || !(sc->node_max[3 * (size_t)i + k] >= sc->node_max[3 * (size_t)c + k])) return false;
    }
    return true;
}

// Every leaf box contains the vertices of its triangles (f32 comparisons on the uploaded arrays).  Together with
// ref_tree_is_monotone this is what pruning (a node's box bounds everything beneath it), the any-hit rule and the blocker
// cache (a leaf whose box is hit is reached by the reference) rest on; every tree classes/bvh.js builds has both properties,
// but the arrays are legal without them and are then traversed literally.
static bool leaves_contain_triangles(const ptx_scene* sc) {
    for (uint32_t i = 0; i < sc->N; i++) {
        if (sc->node_first[i] == -2) continue;
        for (int32_t q = 0; q < sc->node_count[i]; q++) {
            const float* p = sc->coords + 9 * (size_t)sc->inorder_tri[(size_t)sc->node_first[i] + q];
            for (int v = 0; v < 3; v++)
                for (int k = 0; k < 3; k++)
                    if (!(sc->node_min[3 * (size_t)i + k] <= p[3 * v + k]) || !(p[3 * v + k] <= sc->node_max[3 * (size_t)i + k])) return false;
        }
    }
    return true;
}

static double half_area(const float* mn, const float* mx) {
    const double dx = (double)mx[0] - mn[0], dy = (double)mx[1] - mn[1], dz = (double)mx[2] - mn[2];
    return dx * dy + dy * dz + dz * dx;
}

struct SahNode { float mn[3], mx[3]; int fc; int left, right, axis; int parent; double q; /* P(box entered | parent entered), see sah_sample */
                 TriCoef coef; /* largest coefficients beneath (provable pruning) */ };
#ifndef PTX_SAH_CONTRACT_BIAS
#define PTX_SAH_CONTRACT_BIAS 1.0   // 0 = keep every node; measured on the headline: 0.8 -> 91.9, 0.95 .. 1.0 -> 86.3, 1.1 -> 88.5, 1.3 -> 92.0 ms
#endif

// top-down SAH over ids[lo, hi): returns the index of the node created in `tree`
static int sah_build(std::vector<SahItem>& items, std::vector<uint32_t>& ids, size_t lo, size_t hi, std::vector<SahNode>& tree) {
    const int me = (int)tree.size();
    tree.push_back(SahNode());
    SahNode nd;
    for (int k = 0; k < 3; k++) { nd.mn[k] = INFINITY; nd.mx[k] = -INFINITY; }
    for (size_t q = lo; q < hi; q++)
        for (int k = 0; k < 3; k++) { nd.mn[k] = std::min(nd.mn[k], items[ids[q]].mn[k]); nd.mx[k] = std::max(nd.mx[k], items[ids[q]].mx[k]); }
    nd.fc = -2; nd.left = nd.right = -1; nd.axis = 0; nd.parent = -1; nd.q = 1.0; nd.coef = {0.f, 0.f};
    if (hi - lo == 1) {
        nd.fc = items[ids[lo]].fc;
        nd.coef = items[ids[lo]].coef;
    } else {
        const size_t n = hi - lo;
        double best = INFINITY; int best_axis = 0; size_t best_k = n / 2;
        std::vector<uint32_t> order(n);
        std::vector<double> right_cost(n + 1);
        for (int ax = 0; ax < 3; ax++) {
            std::copy(ids.begin() + lo, ids.begin() + hi, order.begin());
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return items[a].cen[ax] < items[b].cen[ax]; });
            float rmn[3] = {INFINITY, INFINITY, INFINITY}, rmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            double rw = 0;
            for (size_t q = n; q-- > 1;) {
                for (int k = 0; k < 3; k++) { rmn[k] = std::min(rmn[k], items[order[q]].mn[k]); rmx[k] = std::max(rmx[k], items[order[q]].mx[k]); }
                rw += items[order[q]].w;
                right_cost[q] = half_area(rmn, rmx) * rw;
            }
            float lmn[3] = {INFINITY, INFINITY, INFINITY}, lmx[3] = {-INFINITY, -INFINITY, -INFINITY};
            double lw = 0;
            for (size_t q = 0; q + 1 < n; q++) {
                for (int k = 0; k < 3; k++) { lmn[k] = std::min(lmn[k], items[order[q]].mn[k]); lmx[k] = std::max(lmx[k], items[order[q]].mx[k]); }
                lw += items[order[q]].w;
                const double cost = half_area(lmn, lmx) * lw + right_cost[q + 1];
                if (cost < best) { best = cost; best_axis = ax; best_k = q + 1; }
            }
        }
        std::stable_sort(ids.begin() + lo, ids.begin() + hi, [&](uint32_t a, uint32_t b) { return items[a].cen[best_axis] < items[b].cen[best_axis]; });
        nd.axis = best_axis;
        nd.left = sah_build(items, ids, lo, lo + best_k, tree);      // lower side along the split axis
        nd.right = sah_build(items, ids, lo + best_k, hi, tree);     // upper side
        tree[nd.left].parent = me; tree[nd.right].parent = me;
        nd.coef = {std::max(tree[nd.left].coef.c_o, tree[nd.right].coef.c_o), std::max(tree[nd.left].coef.c_c, tree[nd.right].coef.c_c)};
    }
    tree[me] = nd;
    return me;
}

// Hit statistics for the contraction below: 2^16 sample rays -- origins uniform on the surfaces, half of them cosine-distributed
// about the normal (bounce rays), half towards a point on a light (shadow rays) -- walked front-to-back through the binary
// tree with the kernels' best-t pruning; q(x) = entered(x) / visited(x), smoothed towards the surface-area ratio where a
// node saw few rays.  Deterministic (fixed seed); affects only the cost of the walk, never its result.
static void sah_sample(const ptx_scene* sc, std::vector<SahNode>& tree) {
    const size_t n = tree.size();
    std::vector<uint32_t> vis(n, 0), ent(n, 0);
    std::vector<double> cum(sc->T);
    double tot = 0;
    auto tri = [&](uint32_t t, int v, int k) { return (double)sc->coords[9 * (size_t)t + 3 * v + k]; };
    for (uint32_t t = 0; t < sc->T; t++) {
        const double a[3] = {tri(t, 1, 0) - tri(t, 0, 0), tri(t, 1, 1) - tri(t, 0, 1), tri(t, 1, 2) - tri(t, 0, 2)};
        const double b[3] = {tri(t, 2, 0) - tri(t, 0, 0), tri(t, 2, 1) - tri(t, 0, 1), tri(t, 2, 2) - tri(t, 0, 2)};
        const double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        tot += 0.5 * std::sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
        cum[t] = tot;
    }
    auto finish = [&]() {
        for (size_t x = 0; x < n; x++) {
            double prior = 1.0;                                    // surface-area ratio, worth 8 pseudo-observations
            if (tree[x].parent >= 0) {
                const double pa = half_area(tree[tree[x].parent].mn, tree[tree[x].parent].mx);
                prior = pa > 0 ? std::min(1.0, half_area(tree[x].mn, tree[x].mx) / pa) : 1.0;
            }
            tree[x].q = ((double)ent[x] + 8.0 * prior) / ((double)vis[x] + 8.0);
        }
    };
    if (!(tot > 0)) { finish(); return; }                          // no area to sample from: surface-area ratios alone
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) * (1.0 / 9007199254740992.0); };
    struct Ray { double o[3], d[3], inv[3]; double best, limit; };
    // recursive walk = the threaded pre-order walk of the octant layout for this ray's direction signs
    std::vector<int> stack;
    for (int r = 0; r < 65536; r++) {
        Ray ray;
        const uint32_t t = (uint32_t)(std::lower_bound(cum.begin(), cum.end(), rnd() * tot) - cum.begin());
        const uint32_t tt = std::min(t, sc->T - 1);
        double u = rnd(), v = rnd();
        if (u + v > 1) { u = 1 - u; v = 1 - v; }
        for (int k = 0; k < 3; k++) ray.o[k] = tri(tt, 0, k) * (1 - u - v) + tri(tt, 1, k) * u + tri(tt, 2, k) * v;
        const double nn[3] = {sc->normals[3 * (size_t)tt], sc->normals[3 * (size_t)tt + 1], sc->normals[3 * (size_t)tt + 2]};
        if ((r & 1) && sc->L > 0) {
            const uint32_t lt = (uint32_t)sc->light_tri[std::min<uint32_t>((uint32_t)(rnd() * sc->L), sc->L - 1)];
            double a = rnd(), b = rnd();
            if (a + b > 1) { a = 1 - a; b = 1 - b; }
            for (int k = 0; k < 3; k++) ray.d[k] = tri(lt, 0, k) * (1 - a - b) + tri(lt, 1, k) * a + tri(lt, 2, k) * b - ray.o[k];
        } else {
            const double th = 6.283185307179586 * rnd(), rr = std::sqrt(rnd()), z = std::sqrt(std::max(0.0, 1 - rr * rr));
            const int ax = std::fabs(nn[0]) < 0.577 ? 0 : (std::fabs(nn[1]) < 0.577 ? 1 : 2);
            double m[3] = {0, 0, 0}; m[ax] = 1;
            double uu[3] = {nn[1] * m[2] - nn[2] * m[1], nn[2] * m[0] - nn[0] * m[2], nn[0] * m[1] - nn[1] * m[0]};
            const double ul = std::sqrt(uu[0] * uu[0] + uu[1] * uu[1] + uu[2] * uu[2]);
            if (!(ul > 0)) continue;
            for (int k = 0; k < 3; k++) uu[k] /= ul;
            const double vv[3] = {nn[1] * uu[2] - nn[2] * uu[1], nn[2] * uu[0] - nn[0] * uu[2], nn[0] * uu[1] - nn[1] * uu[0]};
            for (int k = 0; k < 3; k++) ray.d[k] = rr * std::cos(th) * uu[k] + rr * std::sin(th) * vv[k] + z * nn[k];
        }
        const double dl = std::sqrt(ray.d[0] * ray.d[0] + ray.d[1] * ray.d[1] + ray.d[2] * ray.d[2]);
        if (!(dl > 0)) continue;
        for (int k = 0; k < 3; k++) { ray.d[k] /= dl; ray.inv[k] = 1.0 / ray.d[k]; }
        ray.best = 1e38; ray.limit = 3e38;
        stack.clear();
        stack.push_back(0);
        while (!stack.empty()) {
            const int x = stack.back();
            stack.pop_back();
            const SahNode& nd = tree[x];
            vis[x]++;
            double tn = -1e300, tf = 1e300;
            for (int k = 0; k < 3; k++) {
                const double t1 = ((double)nd.mn[k] - ray.o[k]) * ray.inv[k], t2 = ((double)nd.mx[k] - ray.o[k]) * ray.inv[k];
                tn = std::max(tn, std::min(t1, t2)); tf = std::min(tf, std::max(t1, t2));
            }
            if (!(tn <= tf && tf > 0) || tn > ray.limit) continue;                   // the default rule (option "prune" = 2)
            ent[x]++;
            if (nd.left >= 0) {
                const bool neg = ray.d[nd.axis] < 0;
                stack.push_back(neg ? nd.left : nd.right);       // second
                stack.push_back(neg ? nd.right : nd.left);       // first (popped next)
            } else if (nd.fc >= 0) {
                const uint32_t first = (uint32_t)nd.fc & 0x00FFFFFFu, cnt = ((uint32_t)nd.fc >> 24) & 0x7Fu;
                for (uint32_t q = 0; q < cnt; q++) {
                    const uint32_t ti = (uint32_t)sc->inorder_tri[first + q];
                    const double e1[3] = {tri(ti, 1, 0) - tri(ti, 0, 0), tri(ti, 1, 1) - tri(ti, 0, 1), tri(ti, 1, 2) - tri(ti, 0, 2)};
                    const double e2[3] = {tri(ti, 2, 0) - tri(ti, 0, 0), tri(ti, 2, 1) - tri(ti, 0, 1), tri(ti, 2, 2) - tri(ti, 0, 2)};
                    const double h[3] = {ray.d[1] * e2[2] - ray.d[2] * e2[1], ray.d[2] * e2[0] - ray.d[0] * e2[2], ray.d[0] * e2[1] - ray.d[1] * e2[0]};
                    const double a = e1[0] * h[0] + e1[1] * h[1] + e1[2] * h[2];
                    if (std::fabs(a) < 1e-4) continue;
                    const double f = 1.0 / a, sx[3] = {ray.o[0] - tri(ti, 0, 0), ray.o[1] - tri(ti, 0, 1), ray.o[2] - tri(ti, 0, 2)};
                    const double uq = f * (sx[0] * h[0] + sx[1] * h[1] + sx[2] * h[2]);
                    if (uq < 0 || uq > 1) continue;
                    const double qv[3] = {sx[1] * e1[2] - sx[2] * e1[1], sx[2] * e1[0] - sx[0] * e1[2], sx[0] * e1[1] - sx[1] * e1[0]};
                    const double vq = f * (ray.d[0] * qv[0] + ray.d[1] * qv[1] + ray.d[2] * qv[2]);
                    if (vq < 0 || uq + vq > 1) continue;
                    const double tq = f * (e2[0] * qv[0] + e2[1] * qv[1] + e2[2] * qv[2]);
                    if (tq > 1e-4 && tq < ray.best) { ray.best = tq; ray.limit = tq * (1.0 + PTX_PRUNE_REL) + PTX_PRUNE_ABS; }
                }
            }
        }
    }
    finish();
}

// Contraction: an inner node whose box is almost always hit when its parent's is costs a test and rejects nothing; leaving
// it out of the pre-order list (its children take its place) is free for the result -- the hierarchy above the leaves does
// not matter, see above -- and saves that test.  sah_plan chooses, for every inner node x given its nearest KEPT ancestor
// `anc`, the cheaper of  keep: 1 + P(x entered | anc entered) * (children given x)  and  drop: children given anc.  The
// probabilities are measured: sah_sample walks the uncontracted tree with sample rays of the kind a path tracer shoots.
struct SahPlan {
    enum { MAXD = 12 };                                            // a run of dropped ancestors is at most this long
    const std::vector<SahNode>* tree;
    double bias;                                                   // > 1 favours keeping a node; 0 keeps every node
    std::vector<float> cost;                                       // (node, levels up to the kept ancestor) -> expected tests, < 0 = open
    std::vector<uint8_t> drop;
    void init(const std::vector<SahNode>* t, double b) { tree = t; bias = b; cost.assign(t->size() * (MAXD + 1), -1.0f); drop.assign(cost.size(), 0); }
};
// dist = levels between x and its nearest kept ancestor `anc` (1 = its parent; the root under "nothing kept above" counts 1)
static double sah_plan(SahPlan& pl, int x, int anc, int dist, bool* dropped = nullptr) {
    const SahNode& nd = (*pl.tree)[x];
    if (nd.left < 0) { if (dropped) *dropped = false; return 1.0; }
    const bool memo = dist <= SahPlan::MAXD;
    const size_t key = (size_t)x * (SahPlan::MAXD + 1) + (size_t)(memo ? dist : 0);
    if (memo && pl.cost[key] >= 0.0f) { if (dropped) *dropped = pl.drop[key] != 0; return pl.cost[key]; }
    double p = 1.0;                                                // P(x entered | anc entered): boxes are nested, so the
    for (int y = x; y != anc && y >= 0; y = (*pl.tree)[y].parent) p *= (*pl.tree)[y].q;   // conditionals multiply along the chain
    const double keep = 1.0 + p * (sah_plan(pl, nd.left, x, 1) + sah_plan(pl, nd.right, x, 1));
    bool d = false;
    double c = keep;
    if (pl.bias > 0 && memo && dist < SahPlan::MAXD) {
        const double gone = sah_plan(pl, nd.left, anc, dist + 1) + sah_plan(pl, nd.right, anc, dist + 1);
        if (gone * pl.bias < keep) { d = true; c = gone; }
    }
    if (memo) { pl.cost[key] = (float)c; pl.drop[key] = d ? 1 : 0; }
    if (dropped) *dropped = d;
    return c;
}

// pre-order flattening of `tree` for rays whose direction signs are `oct` (bit k set = d[k] < 0): the child on the
// side the ray comes from is visited first, so the threaded walk is roughly front-to-back and best-t pruning bites
// early.  Which nodes are visited WITHOUT pruning does not depend on the order, so the hit leaves are unchanged.
static void sah_flatten(const std::vector<SahNode>& tree, SahPlan& pl, int node, int anc, int dist, int oct, std::vector<float4>& out, size_t base) {
    auto bits = [](int32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    const SahNode& nd = tree[node];
    const bool neg = (oct >> nd.axis) & 1;
    bool dropped = false;
    sah_plan(pl, node, anc, dist, &dropped);
    if (dropped) {                                                 // its children take its place
        sah_flatten(tree, pl, neg ? nd.right : nd.left, anc, dist + 1, oct, out, base);
        sah_flatten(tree, pl, neg ? nd.left : nd.right, anc, dist + 1, oct, out, base);
        return;
    }
    const size_t me = out.size() / 2 - base;
    out.push_back(make_float4(nd.mn[0], nd.mn[1], nd.mn[2], 0.f));
    // leaves (and empty leaves, -2) keep fc; an inner node carries its packed coefficients under a set sign bit, so every "fc < 0" test holds
    out.push_back(make_float4(nd.mx[0], nd.mx[1], nd.mx[2], bits(nd.left >= 0 ? (int32_t)(0x80000000u | pack_coefs(nd.coef)) : nd.fc)));
    if (nd.left >= 0) {
        sah_flatten(tree, pl, neg ? nd.right : nd.left, node, 1, oct, out, base);
        sah_flatten(tree, pl, neg ? nd.left : nd.right, node, 1, oct, out, base);
    }
    out[2 * (base + me)].w = bits((int32_t)(out.size() / 2 - base));   // first pre-order index after this subtree (fixed below)
}

// out: 8 layouts (one per direction octant) of N2 nodes each, same 32-B record format as the reference tree
static bool build_leaf_sah(const ptx_scene* sc, const std::vector<TriCoef>& leaf_coef, std::vector<float4>& out, int* n2) {
    if (!ref_tree_is_monotone(sc)) return false;
    std::vector<SahItem> items;
    for (uint32_t i = 0; i < sc->N; i++) {
        if (sc->node_first[i] == -2) continue;
        SahItem it;
        for (int k = 0; k < 3; k++) {
            it.mn[k] = sc->node_min[3 * (size_t)i + k]; it.mx[k] = sc->node_max[3 * (size_t)i + k];
            it.cen[k] = 0.5f * it.mn[k] + 0.5f * it.mx[k];
        }
        it.fc = sc->node_count[i] == 0 ? -2 : (sc->node_first[i] | (sc->node_count[i] << 24));
        it.w = std::max(1, sc->node_count[i]);
        it.coef = leaf_coef[i];
        items.push_back(it);
    }
    if (items.empty()) return false;
    // the kernels address node records with 32-bit byte offsets: 8 layouts x (2 * leaves - 1) nodes x 32 B must stay below 4 GiB
    if ((uint64_t)items.size() * 2u * 32u * 8u >= (1ull << 32)) return false;
    std::vector<uint32_t> ids(items.size());
    for (uint32_t q = 0; q < ids.size(); q++) ids[q] = q;
    std::vector<SahNode> tree;
    tree.reserve(items.size() * 2);
    sah_build(items, ids, 0, ids.size(), tree);
    sah_sample(sc, tree);
    SahPlan pl;
    pl.init(&tree, PTX_SAH_CONTRACT_BIAS);
    sah_plan(pl, 0, -1, 1);
    out.clear();
    out.reserve(tree.size() * 16);
    sah_flatten(tree, pl, 0, -1, 1, 0, out, 0);
    const int32_t N2 = (int32_t)(out.size() / 2);                // nodes kept (the same set in every octant layout)
    out.clear();
    auto bits = [](int32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    for (int oct = 0; oct < 8; oct++) {
        const size_t base = (size_t)oct * N2;
        sah_flatten(tree, pl, 0, -1, 1, oct, out, base);
        for (int32_t i = 0; i < N2; i++) {
            int32_t m; std::memcpy(&m, &out[2 * (base + i)].w, 4);
            if (m >= N2) out[2 * (base + i)].w = bits(-1);
            // Near plane first: in the layout of an octant with d[k] < 0 the two bounds of axis k trade places, so a ray of
            // that octant finds its entry plane in the "min" position and its exit plane in the "max" position.  The slab
            // test's min(t1, t2) / max(t1, t2) are symmetric in the two planes, so every kernel reads these records unchanged;
            // pipeline 2 uses the order to drop the six min/max (slab_node_ordered).
            float4& lo = out[2 * (base + i)];
            float4& hi = out[2 * (base + i) + 1];
            if (oct & 1) std::swap(lo.x, hi.x);
            if (oct & 2) std::swap(lo.y, hi.y);
            if (oct & 4) std::swap(lo.z, hi.z);
        }
    }
    *n2 = N2;
    return true;
}

// {min.xyz, miss}, {max.xyz, fc}  ->  {min.x, min.y, max.x, max.y}, {min.z, max.z, miss, fc}   (ptx_kernels.h)
static void repack_nodes(std::vector<float4>& n) {
    for (size_t i = 0; i + 1 < n.size(); i += 2) {
        const float4 a = n[i], b = n[i + 1];
        n[i] = make_float4(a.x, a.y, b.x, b.y);
        n[i + 1] = make_float4(a.z, b.z, a.w, b.w);
    }
}

extern "C" int ptx_upload_scene(ptx_ctx* ctx, const ptx_scene* sc) {
    if (!ctx) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_upload_scene(ctx, sc));
    if (!sc) return fail(ctx, PTX_ERR_INVALID, "ptx_upload_scene: scene is NULL");
    const uint32_t T = sc->T, N = sc->N, L = sc->L;
    if (T == 0 || N == 0) return fail(ctx, PTX_ERR_INVALID, "ptx_upload_scene: T and N must be > 0");
    if (T >= (1u << 24)) return fail(ctx, PTX_ERR_INVALID, "ptx_upload_scene: T must be < 2^24 (the reference's float-encoded indices share this limit)");
    if (!sc->coords || !sc->normals || !sc->colors || !sc->emissions || !sc->node_min || !sc->node_max || !sc->node_miss ||
        !sc->node_first || !sc->node_count || !sc->inorder_tri || (L > 0 && !sc->light_tri))
        return fail(ctx, PTX_ERR_INVALID, "ptx_upload_scene: NULL array");
    // ---- validate every index before it can reach a kernel (a bad miss link would never terminate)
    for (uint32_t i = 0; i < N; i++) {
        const int32_t m = sc->node_miss[i];
        if (!(m == -1 || (m > (int32_t)i && m < (int32_t)N))) return fail(ctx, PTX_ERR_INVALID, "ptx_upload_scene: node_miss[" + std::to_string(i) + "] must be -1 or a later node index");
        const int32_t f = sc->node_first[i], cnt = sc->node_count[i];
        if (f == -2) continue;
        if (f < 0 || cnt < 0 || cnt > 127 || (uint64_t)f + (uint64_t)cnt > T)
            return fail(ctx, PTX_ERR_INVALID, "ptx_upload_scene: leaf range of node " + std::to_string(i) + " is out of bounds (count must be <= 127)");
    }
    for (uint32_t k = 0; k < T; k++)
        if (sc->inorder_tri[k] < 0 || (uint32_t)sc->inorder_tri[k] >= T) return fail(ctx, PTX_ERR_INVALID, "ptx_upload_scene: inorder_tri out of range");
    for (uint32_t l = 0; l < L; l++)
        if (sc->light_tri[l] < 0 || (uint32_t)sc->light_tri[l] >= T) return fail(ctx, PTX_ERR_INVALID, "ptx_upload_scene: light_tri out of range");

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_all(ctx));
    free_scene(ctx);

    std::vector<float4> nodes((size_t)N * 2), tris((size_t)T * 3), pairs((size_t)T * 5), aN(T), aC(T), aE(T), light((size_t)L * 6);
    auto bits = [](int32_t v) { float f; std::memcpy(&f, &v, 4); return f; };
    for (uint32_t i = 0; i < N; i++) {
        const float* mn = sc->node_min + 3 * (size_t)i;
        const float* mx = sc->node_max + 3 * (size_t)i;
        const int32_t f = sc->node_first[i];
        const int32_t fc = (f == -2 || sc->node_count[i] == 0) ? -2 : (f | (sc->node_count[i] << 24));   // an empty leaf walks like an inner node
        nodes[2 * (size_t)i] = make_float4(mn[0], mn[1], mn[2], bits(sc->node_miss[i]));
        nodes[2 * (size_t)i + 1] = make_float4(mx[0], mx[1], mx[2], bits(fc));
    }
    for (uint32_t k = 0; k < T; k++) {
        const int32_t tri = sc->inorder_tri[k];
        const float* p = sc->coords + 9 * (size_t)tri;
        tris[3 * (size_t)k] = make_float4(p[0], p[1], p[2], bits(tri));
        tris[3 * (size_t)k + 1] = make_float4(p[3] - p[0], p[4] - p[1], p[5] - p[2], 0.f); // glsl:151
        tris[3 * (size_t)k + 2] = make_float4(p[6] - p[0], p[7] - p[1], p[8] - p[2], 0.f); // glsl:152
    }
    // ---- provable pruning: scene bounds, then one position-error coefficient per triangle, leaf and inner node (prune_coef)
    float bmn[3] = {INFINITY, INFINITY, INFINITY}, bmx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (size_t q = 0; q < (size_t)T * 9; q++) { bmn[q % 3] = std::min(bmn[q % 3], sc->coords[q]); bmx[q % 3] = std::max(bmx[q % 3], sc->coords[q]); }
    float centre[3];
    double diag2 = 0;
    for (int q = 0; q < 3; q++) { centre[q] = 0.5f * bmn[q] + 0.5f * bmx[q]; diag2 += ((double)bmx[q] - bmn[q]) * ((double)bmx[q] - bmn[q]); ctx->bbox_min[q] = bmn[q]; ctx->bbox_max[q] = bmx[q]; }
    const float two_rs = std::nextafter((float)std::sqrt(diag2), INFINITY) * 1.0001f;    // >= 2 x half-diagonal, also after the centre's rounding
    const float rs = 0.5f * two_rs;
    const float m_cap = 5.001f * rs;                                                  // rays from within |o - centre| <= 4 Rs are prunable (ray_m <= 4.0004 Rs + Rs)
    float e_max = 0.f;                                                                // the longest triangle edge, rounded up (ray_m)
    for (uint32_t k = 0; k < T; k++) {
        const float4 e1 = tris[3 * (size_t)k + 1], e2 = tris[3 * (size_t)k + 2];
        const double l1 = std::sqrt((double)e1.x * e1.x + (double)e1.y * e1.y + (double)e1.z * e1.z), l2 = std::sqrt((double)e2.x * e2.x + (double)e2.y * e2.y + (double)e2.z * e2.z);
        const double dx = (double)e2.x - e1.x, dy = (double)e2.y - e1.y, dz = (double)e2.z - e1.z;
        e_max = std::max(e_max, std::nextafter((float)std::max(std::max(l1, l2), std::sqrt(dx * dx + dy * dy + dz * dz)), INFINITY));
    }
    // classes: up to PTX_MAX_CLS plane normals, each time the normal of the unclassified triangle with the largest c_o
    // among those that may be class members and gain from it; members = the triangles within delta of +- that normal
    std::vector<TriCoef> slot_coef(T);
    V3 cls[PTX_MAX_CLS];
    int n_cls = 0;
    {
        std::vector<double> nx(T), ny(T), nz(T);
        std::vector<float> co(T), cc(T);
        std::vector<uint8_t> open(T, 0);
        for (uint32_t k = 0; k < T; k++) {
            const float4 e1 = tris[3 * (size_t)k + 1], e2 = tris[3 * (size_t)k + 2];
            const double l1 = std::sqrt((double)e1.x * e1.x + (double)e1.y * e1.y + (double)e1.z * e1.z), l2 = std::sqrt((double)e2.x * e2.x + (double)e2.y * e2.y + (double)e2.z * e2.z);
            const double gx = (double)e2.y * e1.z - (double)e1.y * e2.z, gy = (double)e2.z * e1.x - (double)e1.z * e2.x, gz = (double)e2.x * e1.y - (double)e1.x * e2.y;
            const double a2 = std::sqrt(gx * gx + gy * gy + gz * gz), E = std::max(l1, l2);
            co[k] = coef_other_of(E, m_cap);
            cc[k] = coef_class_of(E, a2, m_cap);
            if (a2 > 0) { nx[k] = gx / a2; ny[k] = gy / a2; nz[k] = gz / a2; }
            open[k] = std::isfinite(cc[k]) && cc[k] < co[k];
            slot_coef[k] = {co[k], 0.f};
        }
        const double lim = 0.5 * (0.999 * (double)PTX_CLS_DELTA) * (0.999 * (double)PTX_CLS_DELTA);   // |n - sN| <= delta  <=>  1 - |n . N| <= delta^2 / 2
        for (; n_cls < PTX_MAX_CLS; n_cls++) {
            int64_t pick = -1;
            for (uint32_t k = 0; k < T; k++)
                if (open[k] && (pick < 0 || co[k] > co[pick])) pick = k;
            if (pick < 0 || !(co[pick] > 4.0f * cc[pick])) break;                     // nothing left that gains a factor of 4
            const double Nx = nx[pick], Ny = ny[pick], Nz = nz[pick];
            cls[n_cls] = {(float)Nx, (float)Ny, (float)Nz};
            const double fx = cls[n_cls].x, fy = cls[n_cls].y, fz = cls[n_cls].z;     // the float class normal is what the kernels see
            for (uint32_t k = 0; k < T; k++)
                if (open[k] && 1.0 - std::fabs(nx[k] * fx + ny[k] * fy + nz[k] * fz) / std::sqrt(fx * fx + fy * fy + fz * fz) <= lim) {
                    slot_coef[k] = {0.f, cc[k]};
                    open[k] = 0;
                }
        }
    }
    std::vector<TriCoef> leaf_coef(N, TriCoef{0.f, 0.f});                             // by node index; 0 for inner nodes
    for (uint32_t i = 0; i < N; i++) {
        if (sc->node_first[i] == -2 || sc->node_count[i] <= 0) continue;
        TriCoef c = {0.f, 0.f};
        for (int32_t q = 0; q < sc->node_count[i]; q++) {
            const TriCoef t = slot_coef[(size_t)sc->node_first[i] + q];
            c = {std::max(c.c_o, t.c_o), std::max(c.c_c, t.c_c)};
        }
        leaf_coef[i] = c;
    }
    {   // pipelines 0 / 1 find a leaf's packed coefficients at its first slot (the largest, should leaves share a first slot)
        std::vector<TriCoef> at_slot(T, TriCoef{0.f, 0.f});
        for (uint32_t i = 0; i < N; i++) {
            if (sc->node_first[i] == -2 || sc->node_count[i] <= 0) continue;
            TriCoef& a = at_slot[(size_t)sc->node_first[i]];
            a = {std::max(a.c_o, leaf_coef[i].c_o), std::max(a.c_c, leaf_coef[i].c_c)};
        }
        for (uint32_t i = 0; i < N; i++)
            if (sc->node_first[i] != -2 && sc->node_count[i] > 0) tris[3 * (size_t)sc->node_first[i] + 1].w = bits((int32_t)pack_coefs(at_slot[(size_t)sc->node_first[i]]));
    }
    {   // reference tree: cutting inner node i skips exactly the nodes (i, miss[i]): their largest coefficients go into fc
        std::vector<float> lo(N), lc(N);
        for (uint32_t i = 0; i < N; i++) { lo[i] = leaf_coef[i].c_o; lc[i] = leaf_coef[i].c_c; }
        const RangeMax rmo(lo), rmc(lc);
        for (uint32_t i = 0; i < N; i++) {
            if (!(sc->node_first[i] == -2)) continue;                                 // leaves and empty leaves keep their fc
            const size_t end = sc->node_miss[i] < 0 ? N : (size_t)sc->node_miss[i];
            const TriCoef c = {rmo.query((size_t)i + 1, end), rmc.query((size_t)i + 1, end)};
            nodes[2 * (size_t)i + 1].w = bits((int32_t)(0x80000000u | pack_coefs(c)));
        }
    }
    // blocker cache of pipeline 2: every slot knows the reference-tree node of its leaf (the last one, should leaves share slots:
    // such scenes are not rendered by pipeline 2)
    for (uint32_t i = 0; i < N; i++) {
        if (sc->node_first[i] == -2) continue;
        for (int32_t q = 0; q < sc->node_count[i]; q++) tris[3 * ((size_t)sc->node_first[i] + q) + 2].w = bits((int32_t)i);
    }
    // ---- any-hit termination of light-visibility rays (ptx_kernels.h light_bound): which slots emit, boxes around them
    std::vector<uint8_t> slot_emits(T, 0);
    uint32_t n_lbox = 0, lcoefs = 0;
    float lbox[PTX_MAX_LBOX][6] = {};
    {
        struct Box { float mn[3], mx[3]; };
        auto join = [](const Box& a, const Box& b) { Box r; for (int q = 0; q < 3; q++) { r.mn[q] = std::min(a.mn[q], b.mn[q]); r.mx[q] = std::max(a.mx[q], b.mx[q]); } return r; };
        auto half_area = [](const Box& b) { const double x = (double)b.mx[0] - b.mn[0], y = (double)b.mx[1] - b.mn[1], z = (double)b.mx[2] - b.mn[2]; return x * y + y * z + z * x; };
        std::vector<Box> boxes;
        TriCoef lc = {0.f, 0.f};
        for (uint32_t k = 0; k < T; k++) {
            const int32_t tri = sc->inorder_tri[k];
            const float* e = sc->emissions + 3 * (size_t)tri;
            if (e[0] == 0.f && e[1] == 0.f && e[2] == 0.f) continue;             // NaN counts as emitting
            slot_emits[k] = 1;
            const float* p = sc->coords + 9 * (size_t)tri;
            Box b = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
            for (int v = 0; v < 3; v++) for (int q = 0; q < 3; q++) { b.mn[q] = std::min(b.mn[q], p[3 * v + q]); b.mx[q] = std::max(b.mx[q], p[3 * v + q]); }
            boxes.push_back(b);
            lc = {std::max(lc.c_o, slot_coef[k].c_o), std::max(lc.c_c, slot_coef[k].c_c)};
        }
        if (boxes.size() > 64) {                     // many emitters: chunks along the longest axis of their bounds first
            Box all = boxes[0];
            for (const Box& b : boxes) all = join(all, b);
            int ax = 0;
            for (int q = 1; q < 3; q++) if (all.mx[q] - all.mn[q] > all.mx[ax] - all.mn[ax]) ax = q;
            std::sort(boxes.begin(), boxes.end(), [&](const Box& a, const Box& b) { return a.mn[ax] + a.mx[ax] < b.mn[ax] + b.mx[ax]; });
            std::vector<Box> chunks;
            for (size_t q = 0; q < 64; q++) {
                const size_t lo = q * boxes.size() / 64, hi = (q + 1) * boxes.size() / 64;
                Box c = boxes[lo];
                for (size_t r = lo + 1; r < hi; r++) c = join(c, boxes[r]);
                chunks.push_back(c);
            }
            boxes.swap(chunks);
        }
        // greedy: join the pair whose union adds the least surface -- down to PTX_MAX_LBOX boxes, and further while the union is
        // no larger than twice its parts (a lamp's triangles, the two halves of a ceiling panel): every box costs each
        // light-visibility ray a slab test (measured on pixar_room, whose 19 emitters are one lamp: 1 / 2 / 4 / 8 / 16 boxes
        // 75.6 / 76.0 / 76.6 / 78.2 / 80.8 ms per frame)
        while (boxes.size() > 1) {
            size_t ba = 0, bb = 1;
            double best = INFINITY;
            for (size_t a = 0; a < boxes.size(); a++)
                for (size_t b = a + 1; b < boxes.size(); b++) {
                    const double g = half_area(join(boxes[a], boxes[b])) - half_area(boxes[a]) - half_area(boxes[b]);
                    if (g < best) { best = g; ba = a; bb = b; }
                }
            if (boxes.size() <= PTX_MAX_LBOX && half_area(join(boxes[ba], boxes[bb])) > 2.0 * (half_area(boxes[ba]) + half_area(boxes[bb]))) break;
            boxes[ba] = join(boxes[ba], boxes[bb]);
            boxes.erase(boxes.begin() + (ptrdiff_t)bb);
        }
        n_lbox = (uint32_t)boxes.size();
        lcoefs = pack_coefs(lc);
        for (uint32_t k = 0; k < n_lbox; k++) {
            // padded: under the margin rule a box the slab test misses is taken to hold no hit (light_bound, PRUNE = 2)
            const Box& b = boxes[k];
            const double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2];
            const float pad = (float)(1.0e-3 * std::sqrt(dx * dx + dy * dy + dz * dz) + 0.5e-3 * (double)two_rs);
            for (int q = 0; q < 3; q++) { lbox[k][q] = b.mn[q] - pad; lbox[k][3 + q] = b.mx[q] + pad; }
        }
    }
    // Pair records live at index first + p, which is only collision-free when the leaves' slot ranges are disjoint (they
    // partition the inorder list in every tree classes/bvh.js builds).  Arrays that share slots between leaves are still
    // legal input: such a scene is rendered by pipeline 1, which reads the 48-B records.
    bool disjoint = true;
    {
        std::vector<uint8_t> used(T, 0);
        for (uint32_t i = 0; i < N && disjoint; i++) {
            if (sc->node_first[i] == -2) continue;
            for (int32_t q = 0; q < sc->node_count[i]; q++) {
                uint8_t& u = used[(size_t)sc->node_first[i] + q];
                if (u) { disjoint = false; break; }
                u = 1;
            }
        }
    }
    for (uint32_t i = 0; i < N && disjoint; i++) {   // pipeline 2: the leaves' triangles again, two per record, component-interleaved
        if (sc->node_first[i] == -2) continue;
        const int32_t first = sc->node_first[i], cnt = sc->node_count[i];
        for (int32_t q = 0; 2 * q < cnt; q++) {
            float v[2][9] = {};
            for (int h = 0; h < 2 && 2 * q + h < cnt; h++) {
                const float4* r = &tris[3 * (size_t)(first + 2 * q + h)];
                const float c9[9] = {r[0].x, r[0].y, r[0].z, r[1].x, r[1].y, r[1].z, r[2].x, r[2].y, r[2].z};
                std::memcpy(v[h], c9, sizeof(c9));
            }
            float4* o = &pairs[5 * (size_t)(first + q)];
            o[0] = make_float4(v[0][0], v[1][0], v[0][1], v[1][1]);
            o[1] = make_float4(v[0][2], v[1][2], v[0][3], v[1][3]);
            o[2] = make_float4(v[0][4], v[1][4], v[0][5], v[1][5]);
            o[3] = make_float4(v[0][6], v[1][6], v[0][7], v[1][7]);
            int32_t rz = cnt - 2 * q;                                                  // + emission flags of A (bit 16) and B (bit 17)
            for (int h = 0; h < 2 && 2 * q + h < cnt; h++) rz |= (int32_t)slot_emits[(size_t)(first + 2 * q + h)] << (16 + h);
            o[4] = make_float4(v[0][8], v[1][8], bits(rz), q == 0 ? bits((int32_t)pack_coefs(leaf_coef[i])) : 0.f);
        }
    }
    for (uint32_t t = 0; t < T; t++) {
        aN[t] = make_float4(sc->normals[3 * (size_t)t], sc->normals[3 * (size_t)t + 1], sc->normals[3 * (size_t)t + 2], 0.f);
        aC[t] = make_float4(sc->colors[3 * (size_t)t], sc->colors[3 * (size_t)t + 1], sc->colors[3 * (size_t)t + 2], 0.f);
        aE[t] = make_float4(sc->emissions[3 * (size_t)t], sc->emissions[3 * (size_t)t + 1], sc->emissions[3 * (size_t)t + 2], 0.f);
    }
    for (uint32_t l = 0; l < L; l++) {
        const int32_t tri = sc->light_tri[l];
        const float* p = sc->coords + 9 * (size_t)tri;
        // glsl:390-392: lightArea = 0.5 * length(cross(v1 - v0, v2 - v0)); pdf = 1 / (float(L) * lightArea)
        const float ax = p[3] - p[0], ay = p[4] - p[1], az = p[5] - p[2];
        const float bx = p[6] - p[0], by = p[7] - p[1], bz = p[8] - p[2];
        const float cx = ay * bz - by * az, cy = az * bx - bz * ax, cz = ax * by - bx * ay;
        const float area = 0.5f * std::sqrt((cx * cx + cy * cy) + cz * cz);
        const float pdf = 1.0f / ((float)L * area);
        light[6 * (size_t)l + 0] = make_float4(p[0], p[1], p[2], 0.f);
        light[6 * (size_t)l + 1] = make_float4(p[3], p[4], p[5], 0.f);
        light[6 * (size_t)l + 2] = make_float4(p[6], p[7], p[8], 0.f);
        light[6 * (size_t)l + 3] = aN[tri];
        light[6 * (size_t)l + 4] = aE[tri];
        light[6 * (size_t)l + 5] = make_float4(pdf, area, 0.f, 0.f);
    }
    // the 256 largest triangles (by area, as inorder slots): what ptx_selfcheck_pruning aims its grazing rays at
    std::vector<uint32_t> big(T);
    {
        std::vector<float> area(T);
        for (uint32_t k = 0; k < T; k++) {
            const float4 e1 = tris[3 * (size_t)k + 1], e2 = tris[3 * (size_t)k + 2];
            const float cx = e1.y * e2.z - e2.y * e1.z, cy = e1.z * e2.x - e2.z * e1.x, cz = e1.x * e2.y - e2.x * e1.y;
            area[k] = cx * cx + cy * cy + cz * cz;
            big[k] = k;
        }
        const size_t keep = std::min<size_t>(256, T);
        std::partial_sort(big.begin(), big.begin() + keep, big.end(), [&](uint32_t a, uint32_t b) { return area[a] > area[b] || (area[a] == area[b] && a < b); });
        big.resize(keep);
    }
    std::vector<uint32_t> emit_slots;
    for (uint32_t k = 0; k < T; k++) if (slot_emits[k]) emit_slots.push_back(k);
    ctx->n_emit_slots = (uint32_t)emit_slots.size();
    if (emit_slots.empty()) emit_slots.push_back(0);            // upload() wants a non-empty buffer
    // the emitter list of the provable any-hit rule (ptx_kernels.h light_bound): G' = (e2 x e1) / n, n = 1.01 * 90 u E^2, in double from
    // the f32 edges the kernels use, rounded once
    std::vector<float4> emit_list;
    double e_min = INFINITY;
    if (ctx->n_emit_slots > 0 && ctx->n_emit_slots <= PTX_MAX_EMIT) {
        for (uint32_t q = 0; q < ctx->n_emit_slots; q++) {
            const uint32_t k = emit_slots[q];
            const float4 e1 = tris[3 * (size_t)k + 1], e2 = tris[3 * (size_t)k + 2];
            const double gx = (double)e2.y * e1.z - (double)e1.y * e2.z, gy = (double)e2.z * e1.x - (double)e1.z * e2.x, gz = (double)e2.x * e1.y - (double)e1.x * e2.y;
            const double l1 = std::sqrt((double)e1.x * e1.x + (double)e1.y * e1.y + (double)e1.z * e1.z), l2 = std::sqrt((double)e2.x * e2.x + (double)e2.y * e2.y + (double)e2.z * e2.z);
            const double E = std::max(l1, l2), n = 1.01 * 90.0 * kU * E * E;
            if (!(E > 0.0) || !std::isfinite(E) || !std::isfinite(gx / n) || !std::isfinite(gy / n) || !std::isfinite(gz / n)) { emit_list.clear(); break; }   // degenerate: no list
            e_min = std::min(e_min, E);
            emit_list.push_back(make_float4((float)(gx / n), (float)(gy / n), (float)(gz / n), 0.f));
        }
    }
    while (!emit_list.empty() && emit_list.size() % 4 != 0) emit_list.push_back(emit_list.back());   // light_bound reads four at a time (a repeated entry changes no minimum)
    const float emit_c1 = emit_list.empty() ? 0.f : std::nextafter((float)(0.672 / e_min), INFINITY);
    ctx->n_emit_list = (uint32_t)emit_list.size();
    ctx->emit_c1 = emit_c1;
    ctx->d_emit_list = nullptr;
    DScene S = {};
    int rc;
    if (!emit_list.empty() && (rc = upload(ctx, emit_list, &ctx->d_emit_list))) { free_scene(ctx); return rc; }
    const std::vector<float4> ref_nodes = nodes;      // {min.xyz, miss}, {max.xyz, fc}: becomes layout 8 behind the octant layouts
    repack_nodes(nodes);
    if ((rc = upload(ctx, nodes, &S.nodes)) || (rc = upload(ctx, tris, &S.tris)) || (rc = upload(ctx, pairs, &S.pairs)) || (rc = upload(ctx, aN, &S.attrN)) ||
        (rc = upload(ctx, aC, &S.attrC)) || (rc = upload(ctx, aE, &S.attrE)) || (rc = upload(ctx, light, &S.light)) ||
        (rc = upload(ctx, big, &ctx->d_big_slots)) || (rc = upload(ctx, emit_slots, &ctx->d_emit_slots))) {
        free_scene(ctx);
        return rc;
    }
    S.N = (int)N; S.T = (int)T; S.L = (int)L;
    S.center = {centre[0], centre[1], centre[2]}; S.rs = rs; S.e_max = e_max; S.m_cap = m_cap;
    S.n_cls = n_cls;
    for (int j = 0; j < PTX_MAX_CLS; j++) S.cls[j] = cls[j < n_cls ? j : 0];      // ray_kappa takes the minimum over all PTX_MAX_CLS entries
    S.n_lbox = n_lbox; S.lcoefs = lcoefs;
    std::memcpy(S.lbox, lbox, sizeof(lbox));
    ctx->nodes_sah = nullptr; ctx->N_sah = 0;
    {
        std::vector<float4> sah;
        int n2 = 0;
        if (build_leaf_sah(sc, leaf_coef, sah, &n2)) {
            // nine layouts of nmax records each: the eight octant layouts of the leaf-SAH tree, then the reference tree
            // itself, which rays with an infinite 1/d component walk (slab_node_ref).  Shorter layouts end in records that
            // no ray enters (NaN box, miss link -1, nothing to queue), so one node count serves all nine.
            const size_t nmax = std::max((size_t)n2, (size_t)N);
            if ((uint64_t)nmax * 32u * 9u < (1ull << 32)) {
                const float nan = std::numeric_limits<float>::quiet_NaN();    // NaN bounds: a miss under every slab test here
                std::vector<float4> all(nmax * 2 * 9);
                for (size_t q = 0; q < nmax * 9; q++) {
                    all[2 * q] = make_float4(nan, nan, nan, bits(-1));
                    all[2 * q + 1] = make_float4(nan, nan, nan, bits(-2));
                }
                for (int oct = 0; oct < 8; oct++)
                    std::copy(sah.begin() + (size_t)oct * n2 * 2, sah.begin() + (size_t)(oct + 1) * n2 * 2, all.begin() + (size_t)oct * nmax * 2);
                std::copy(ref_nodes.begin(), ref_nodes.end(), all.begin() + (size_t)8 * nmax * 2);
                repack_nodes(all);
                if ((rc = upload(ctx, all, &ctx->nodes_sah))) { free_scene(ctx); return rc; }
                ctx->N_sah = (int)nmax;
            }
        }
    }
    ctx->S = S;
    ctx->n_big_slots = (uint32_t)big.size();
    ctx->pool_ok = disjoint;
    ctx->prune_ok = ref_tree_is_monotone(sc) && leaves_contain_triangles(sc);
    ctx->have_scene = true;
    ctx->prim_valid = false;
    return PTX_OK;
}

extern "C" int ptx_reset_accum(ptx_ctx* ctx) {
    if (!ctx) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_for_all(ctx, ptx_reset_accum));
    if (!ctx->have_params) return fail(ctx, PTX_ERR_STATE, "ptx_reset_accum: ptx_set_params first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_all(ctx));                                  // nothing of either launch slot may still be adding to it
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_accum, 0, (size_t)ctx->P.W * ctx->P.H * sizeof(float4), ctx->stream));
    for (int k = 0; k < 2; k++)
        if (ctx->d_compat[k]) HIP_TRY(ctx, hipMemsetAsync(ctx->d_compat[k], 0, (size_t)ctx->P.W * ctx->P.H * sizeof(uchar4), ctx->stream));
    ctx->frames_done = 0;
    return PTX_OK;
}

extern "C" int ptx_set_params(ptx_ctx* ctx, const ptx_params* p) {
    if (!ctx) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_set_params(ctx, p));
    if (!p) return fail(ctx, PTX_ERR_INVALID, "ptx_set_params: params is NULL");
    if (p->width == 0 || p->height == 0 || (uint64_t)p->width * p->height > (1ull << 26)) return fail(ctx, PTX_ERR_INVALID, "ptx_set_params: bad image size (width * height must be in 1 .. 2^26)");
    if (p->sample_count < 1) return fail(ctx, PTX_ERR_INVALID, "ptx_set_params: sample_count must be >= 1");
    if (p->max_path_length > 65535) return fail(ctx, PTX_ERR_INVALID, "ptx_set_params: max_path_length must be <= 65535");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_all(ctx));
    const bool resize = !ctx->have_params || ctx->P.W != p->width || ctx->P.H != p->height;
    DParams P = {};
    P.W = p->width; P.H = p->height;
    P.src = {p->cam_src[0], p->cam_src[1], p->cam_src[2]};
    P.up = {p->cam_up[0], p->cam_up[1], p->cam_up[2]};
    P.right = {p->cam_right[0], p->cam_right[1], p->cam_right[2]};
    P.lb = {p->cam_left_bottom[0], p->cam_left_bottom[1], p->cam_left_bottom[2]};
    P.aspect = (float)((double)p->width / (double)p->height); // main.js:291
    P.max_len = p->max_path_length;
    P.spp = p->sample_count;
    if (resize) {
        free_params_buffers(ctx);
        const size_t npix = (size_t)P.W * P.H;
        HIP_TRY(ctx, hipMalloc(&ctx->d_prim_t, npix * sizeof(float)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_prim_tri, npix * sizeof(int)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_accum, npix * sizeof(float4)));
    }
    ctx->P = P;
    ctx->have_params = true;
    ctx->prim_valid = false;
    if (resize) return ptx_reset_accum(ctx);
    return PTX_OK;
}

extern "C" int ptx_set_option(ptx_ctx* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_set_option(ctx, key, value));
    const std::string k(key);
    if (k == "traversal" && value >= 0 && value <= 3) { ctx->opt_traversal = (int)value; ctx->prim_valid = false; }
    else if (k == "prune" && (value == 1 || value == 2)) { ctx->opt_prune = (int)value; ctx->prim_valid = false; }
    else if (k == "overlap" && (value == 0 || value == 1)) ctx->opt_overlap = (int)value;
    else if (k == "anyhit" && (value == 0 || value == 1)) ctx->opt_anyhit = (int)value;
    else if (k == "emit_list" && (value == 0 || value == 1)) ctx->opt_emit_list = (int)value;
    else if (k == "margin_lat" && value >= 0 && value <= 1000000) ctx->opt_margin_lat = (int)value;
    else if (k == "memo" && (value == 0 || value == 1)) ctx->opt_memo = (int)value;
    else if (k == "count" && (value == 0 || value == 1)) ctx->opt_count = (int)value;
    else if (k == "pipeline" && (value >= 0 && value <= 2)) ctx->opt_pipeline = (int)value;
    else if (k == "pool_chunk" && value >= 1 && value <= 65536) ctx->opt_pool_chunk = (int)value;
    else if (k == "compat_rgba8" && value >= 0 && value <= 2) ctx->opt_compat = (int)value;
    else if (k == "frame_batch" && value >= 0 && value <= 4096) ctx->opt_frame_batch = (int)value;
    else if (k == "tasks_per_wave" && value >= 64 && value <= (1 << 20)) ctx->opt_tasks_per_wave = (int)value;
    else return fail(ctx, PTX_ERR_INVALID, "ptx_set_option: unknown key or bad value: " + k);
    return PTX_OK;
}

static int ensure_primary(ptx_ctx* ctx, float4* d_dirs) {
    if (!ctx->have_scene || !ctx->have_params) return fail(ctx, PTX_ERR_STATE, "scene and params must be set first");
    if (ctx->prim_valid && !d_dirs) return PTX_OK;
    const uint32_t tiles = ((ctx->P.W + 7) / 8) * ((ctx->P.H + 7) / 8);
    // primary rays are always traced WITHOUT pruning (2 M rays once per camera: 0.3 ms): the per-pixel primary-hit cache, and
    // with it every primary hit of every frame, is the unpruned closest hit whatever option "prune" says
    HIP_TRY(ctx, sync_all(ctx));                                // a path tracer still running on either launch slot reads the cache
    hipLaunchKernelGGL(k_primary<0>, dim3(tiles), dim3(64), 0, ctx->stream, scene_view(ctx), ctx->P, ctx->d_prim_t, ctx->d_prim_tri, d_dirs);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));            // once per camera: both launch slots read the cache
    ctx->prim_valid = true;
    return PTX_OK;
}

static int get_event(ptx_ctx* ctx, hipEvent_t* out) {
    if (!ctx->free_events.empty()) { *out = ctx->free_events.back(); ctx->free_events.pop_back(); return PTX_OK; }
    HIP_TRY(ctx, hipEventCreate(out));
    return PTX_OK;
}

static void drain_events(ptx_ctx* ctx) { // stream must be idle
    for (auto& pr : ctx->pending_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) ctx->kernel_ms += ms;
        ctx->free_events.push_back(pr.first);
        ctx->free_events.push_back(pr.second);
    }
    ctx->pending_events.clear();
}

// The device buffers one batch of frames renders through: the library's own allocations (ptx_render_frames) or pieces of
// the static scratch array (ptx_render_frames_noalloc).
struct RenderBufs {
    float4* fb;               // nf x W*H clamp(mean) images of the batch
    const int32_t* ts;        // nf timestamps of the batch
    uint32_t* task_counter;   // pipeline 2: next undrawn task of the running launch
    hipStream_t stream;       // the launch slot's stream
    hipEvent_t wait_before_sum;   // the other slot's last accumulate (frame order of the sum), or nullptr
};

template <int PRUNE, bool MEMO, bool COUNT>
static hipError_t launch_render(ptx_ctx* ctx, const RenderBufs& rb, uint32_t nf) {
    const uint32_t tiles = ((ctx->P.W + 7) / 8) * ((ctx->P.H + 7) / 8);
    hipLaunchKernelGGL((k_render<PRUNE, MEMO, COUNT>), dim3(tiles, nf), dim3(64), 0, rb.stream, scene_view(ctx), ctx->P, rb.ts, ctx->d_prim_t,
                       ctx->d_prim_tri, rb.fb, ctx->d_counters);
    return hipGetLastError();
}

template <int PRUNE, bool MEMO, bool COUNT>
static hipError_t launch_render_wave(ptx_ctx* ctx, const RenderBufs& rb, uint32_t nf) {
    const uint32_t tiles = ((ctx->P.W + 7) / 8) * ((ctx->P.H + 7) / 8);
    const uint64_t total = (uint64_t)tiles * 64u * nf;   // < 2^32: render_frames_impl caps the batch
    const uint32_t per_wave = (uint32_t)ctx->opt_tasks_per_wave;
    const uint32_t blocks = (uint32_t)((total + per_wave - 1) / per_wave);
    hipLaunchKernelGGL((k_render_wave<PRUNE, MEMO, COUNT>), dim3(blocks), dim3(64), 0, rb.stream, scene_view(ctx), ctx->P, rb.ts, ctx->d_prim_t,
                       ctx->d_prim_tri, rb.fb, nf, per_wave, ctx->d_counters);
    return hipGetLastError();
}

template <int PRUNE, bool MEMO, bool COUNT>
static hipError_t launch_render_pool(ptx_ctx* ctx, const RenderBufs& rb, uint32_t nf) {
    const uint32_t tiles = ((ctx->P.W + 7) / 8) * ((ctx->P.H + 7) / 8);
    const uint64_t total = (uint64_t)tiles * 64u * nf;   // < 2^32 - 2^28: render_frames_impl caps the batch
    const uint32_t chunk = (uint32_t)ctx->opt_pool_chunk;
    uint64_t& resident = ctx->pool_resident[PRUNE * 4 + (MEMO ? 2 : 0) + (COUNT ? 1 : 0)];   // waves one residency of the chip holds
    if (resident == 0) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render_pool<PRUNE, MEMO, COUNT>, 64, 0) != hipSuccess || per_cu < 1) per_cu = 16;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) != hipSuccess || cus < 1) cus = 256;
        resident = (uint64_t)per_cu * (uint64_t)cus;   // any value is correct (waves draw work until none is left); this one fills the chip
    }
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(resident, (total + chunk - 1) / chunk);
    hipError_t e = hipMemsetAsync(rb.task_counter, 0, sizeof(uint32_t), rb.stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_render_pool<PRUNE, MEMO, COUNT>), dim3(blocks), dim3(64), 0, rb.stream, scene_view(ctx), ctx->P, rb.ts, ctx->d_prim_t,
                       ctx->d_prim_tri, rb.fb, nf, chunk, rb.task_counter, ctx->d_counters);
    return hipGetLastError();
}

// the path-tracing kernel of the selected pipeline / traversal / memo / count options over one batch
static hipError_t launch_path_tracer(ptx_ctx* ctx, const RenderBufs& rb, uint32_t nf) {
    const int sel = prune_mode(ctx) * 4 + (ctx->opt_memo ? 2 : 0) + (ctx->opt_count ? 1 : 0);
    const int pipeline = (ctx->opt_pipeline == 2 && !ctx->pool_ok) ? 1 : ctx->opt_pipeline;
#define PTX_SEL(fn)                                                   \
    switch (sel) {                                                    \
        case 0: return fn<0, false, false>(ctx, rb, nf);              \
        case 1: return fn<0, false, true>(ctx, rb, nf);               \
        case 2: return fn<0, true, false>(ctx, rb, nf);               \
        case 3: return fn<0, true, true>(ctx, rb, nf);                \
        case 4: return fn<1, false, false>(ctx, rb, nf);              \
        case 5: return fn<1, false, true>(ctx, rb, nf);               \
        case 6: return fn<1, true, false>(ctx, rb, nf);               \
        case 7: return fn<1, true, true>(ctx, rb, nf);                \
        case 8: return fn<2, false, false>(ctx, rb, nf);              \
        case 9: return fn<2, false, true>(ctx, rb, nf);               \
        case 10: return fn<2, true, false>(ctx, rb, nf);              \
        default: return fn<2, true, true>(ctx, rb, nf);               \
    }
    if (pipeline == 2) { PTX_SEL(launch_render_pool) }
    if (pipeline == 1) { PTX_SEL(launch_render_wave) }
    PTX_SEL(launch_render)
#undef PTX_SEL
}

// frames one launch may hold: pipelines 1 and 2 keep the frame-in-batch in 6 bits and index (pixel, frame) tasks with 32 bits
static size_t cap_batch(const ptx_ctx* ctx, size_t batch) {
    if (ctx->opt_pipeline != 0) batch = std::min<size_t>(batch, 64);
    const uint64_t padded = (uint64_t)((ctx->P.W + 7) / 8) * ((ctx->P.H + 7) / 8) * 64u;
    return std::max<size_t>(1, std::min<size_t>(batch, (size_t)(0xE0000000ull / padded)));
}

// one batch: path tracer into rb.fb, frame-order sum into the accumulation, optional RGBA8 compat blend
static int render_batch(ptx_ctx* ctx, const RenderBufs& rb, uint32_t nf, uint32_t frame_number0, bool timed) {
    const size_t npix = (size_t)ctx->P.W * ctx->P.H;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
        if (ctx->pending_events.size() >= 64) {   // a caller that never synchronises: keep the event list bounded
            HIP_TRY(ctx, sync_all(ctx));
            drain_events(ctx);
        }
        int rc = get_event(ctx, &e0);
        if (rc == PTX_OK && (rc = get_event(ctx, &e1)) != PTX_OK) ctx->free_events.push_back(e0);
        if (rc) return rc;
        HIP_TRY(ctx, hipEventRecord(e0, rb.stream));
    }
    HIP_TRY(ctx, launch_path_tracer(ctx, rb, nf));
    if (timed) {
        HIP_TRY(ctx, hipEventRecord(e1, rb.stream));
        ctx->pending_events.emplace_back(e0, e1);
    }
    ctx->kernel_launches++;
    if (rb.wait_before_sum) HIP_TRY(ctx, hipStreamWaitEvent(rb.stream, rb.wait_before_sum, 0));   // frames are summed in order
    hipLaunchKernelGGL(k_accumulate, dim3((uint32_t)((npix + 255) / 256)), dim3(256), 0, rb.stream, ctx->d_accum, rb.fb, (uint32_t)npix, nf);
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->opt_compat) {   // the reference's own progressive blend, one frame after the other through the 8-bit targets
        for (uint32_t f = 0; f < nf; f++) {
            const float bf = 1.0f / (float)(frame_number0 + f + 1u);        // glsl:537: 1.0 / float(frameNumber + 1)
            hipLaunchKernelGGL(k_blend_rgba8, dim3((uint32_t)((npix + 255) / 256)), dim3(256), 0, rb.stream, ctx->d_compat[ctx->compat_cur],
                               rb.fb + (size_t)f * npix, ctx->d_compat[ctx->compat_cur ^ 1], ctx->P.W, ctx->P.H, ctx->P.aspect, bf, ctx->opt_compat);
            ctx->compat_cur ^= 1;
        }
        HIP_TRY(ctx, hipGetLastError());
    }
    ctx->frames_done += nf;
    return PTX_OK;
}

extern "C" int ptx_render_frames(ptx_ctx* ctx, const int32_t* timestamps, uint32_t n_frames, uint32_t first_frame_number) {
    if (!ctx) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_render_frames(ctx, timestamps, n_frames, first_frame_number, false));
    if (!timestamps && n_frames) return fail(ctx, PTX_ERR_INVALID, "ptx_render_frames: timestamps is NULL");
    if (!ctx->have_scene || !ctx->have_params) return fail(ctx, PTX_ERR_STATE, "ptx_render_frames: scene and params must be set first");
    if (n_frames == 0) return PTX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_primary(ctx, nullptr);
    if (rc) return rc;

    const size_t npix = (size_t)ctx->P.W * ctx->P.H;
    // frames rendered concurrently: bounded by an 8 GiB frame-buffer budget (HBM is 288 GB) and 65535 grid.y
    size_t batch = ctx->opt_frame_batch > 0 ? (size_t)ctx->opt_frame_batch : std::min<size_t>(64, std::max<size_t>(1, (8ull << 30) / (npix * sizeof(float4))));
    batch = cap_batch(ctx, std::min<size_t>(batch, n_frames));
    // this call's launch slot: the two alternate, so its kernel overlaps the drain of the previous call's (LaunchSlot)
    LaunchSlot& sl = ctx->slot[ctx->opt_overlap ? ctx->next_slot : 0];
    LaunchSlot& other = ctx->slot[&sl == &ctx->slot[0] ? 1 : 0];
    if (sl.fb_frames < batch) {
        HIP_TRY(ctx, hipStreamSynchronize(sl.stream));
        if (sl.d_fb) (void)hipFree(sl.d_fb);
        sl.d_fb = nullptr; sl.fb_frames = 0;
        HIP_TRY(ctx, hipMalloc(&sl.d_fb, batch * npix * sizeof(float4)));
        sl.fb_frames = batch;
    }
    if (sl.ts_cap < n_frames) {
        HIP_TRY(ctx, hipStreamSynchronize(sl.stream));
        if (sl.d_ts) (void)hipFree(sl.d_ts);
        sl.d_ts = nullptr; sl.ts_cap = 0;
        HIP_TRY(ctx, hipMalloc(&sl.d_ts, (size_t)n_frames * sizeof(int32_t)));
        sl.ts_cap = n_frames;
    }
    if (ctx->opt_compat)
        for (int k = 0; k < 2; k++)
            if (!ctx->d_compat[k]) {
                HIP_TRY(ctx, sync_all(ctx));
                HIP_TRY(ctx, hipMalloc(&ctx->d_compat[k], npix * sizeof(uchar4)));
                HIP_TRY(ctx, hipMemset(ctx->d_compat[k], 0, npix * sizeof(uchar4)));
            }
    // the timestamps travel by value in kernel arguments, 64 per launch: no copy from pageable host memory, which would make
    // the call wait for everything queued before it; the caller's array is free to go when this returns
    for (uint32_t f0 = 0; f0 < n_frames; f0 += PTX_SCRATCH_MAX_BATCH) {
        const uint32_t nf = std::min<uint32_t>(PTX_SCRATCH_MAX_BATCH, n_frames - f0);
        TsBlock tb = {};
        std::memcpy(tb.v, timestamps + f0, nf * sizeof(int32_t));
        hipLaunchKernelGGL(k_store_timestamps, dim3(1), dim3(64), 0, sl.stream, tb, sl.d_ts + f0, nf);
    }
    HIP_TRY(ctx, hipGetLastError());
    for (uint32_t f0 = 0; f0 < n_frames; f0 += (uint32_t)batch) {
        const uint32_t nf = (uint32_t)std::min<size_t>(batch, n_frames - f0);
        const RenderBufs rb = {sl.d_fb, sl.d_ts + f0, sl.d_task_counter, sl.stream, other.busy ? other.ev_done : nullptr};
        if ((rc = render_batch(ctx, rb, nf, first_frame_number + f0, true))) return rc;
    }
    HIP_TRY(ctx, hipEventRecord(sl.ev_done, sl.stream));
    sl.busy = true;
    if (ctx->opt_overlap) ctx->next_slot ^= 1;
    return PTX_OK;
}

// The same rendering with every intermediate in the static scratch: no hipMalloc / hipFree / event creation, no copy from
// host memory (timestamps go by value into k_store_timestamps).  Shares render_batch and all kernels with
// ptx_render_frames.  All contexts on one GPU share that GPU's scratch: calls that overlap on one GPU race.
extern "C" int ptx_render_frames_noalloc(ptx_ctx* ctx, const int32_t* timestamps, uint32_t n_frames, uint32_t first_frame_number) {
    if (!ctx) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_render_frames(ctx, timestamps, n_frames, first_frame_number, true));
    if (!timestamps && n_frames) return fail(ctx, PTX_ERR_INVALID, "ptx_render_frames_noalloc: timestamps is NULL");
    if (!ctx->have_scene || !ctx->have_params) return fail(ctx, PTX_ERR_STATE, "ptx_render_frames_noalloc: scene and params must be set first");
    if (!ctx->d_scratch) return fail(ctx, PTX_ERR_STATE, "ptx_render_frames_noalloc: no scratch on this device");
    const size_t npix = (size_t)ctx->P.W * ctx->P.H;
    const size_t cap = (size_t)PTX_SCRATCH_FB_BYTES / sizeof(float4);        // (pixel, frame) results the scratch holds
    if (npix > cap)
        return fail(ctx, PTX_ERR_INVALID, "ptx_render_frames_noalloc: width * height = " + std::to_string(npix) + " exceeds the static scratch (" +
                                              std::to_string(cap) + " pixels = 64 MiB of frame buffer); use ptx_render_frames");
    if (ctx->opt_compat && (!ctx->d_compat[0] || !ctx->d_compat[1]))
        return fail(ctx, PTX_ERR_STATE, "ptx_render_frames_noalloc: option compat_rgba8 needs its 8-bit targets, which only ptx_render_frames allocates");
    if (n_frames == 0) return PTX_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_primary(ctx, nullptr);                                    // allocates nothing; synchronises once per camera
    if (rc) return rc;
    if (!scratch_claim(ctx))
        return fail(ctx, PTX_ERR_STATE, "ptx_render_frames_noalloc: the static scratch of device " + std::to_string(ctx->device) +
                                            " is in use by another context; ptx_synchronize that context first");
    unsigned char* const base = ctx->d_scratch;                               // the device address, never the symbol's name
    int32_t* const d_ts = (int32_t*)(base + PTX_SCRATCH_TS_OFFSET);
    LaunchSlot& sl = ctx->slot[0];                                            // one scratch per GPU: always slot 0, behind slot 1's work
    LaunchSlot& other = ctx->slot[1];
    if (other.busy) HIP_TRY(ctx, hipStreamWaitEvent(sl.stream, other.ev_done, 0));
    const RenderBufs rb = {(float4*)(base + PTX_SCRATCH_HEADER), d_ts, (uint32_t*)base, sl.stream, nullptr};
    const size_t batch = cap_batch(ctx, std::min<size_t>(std::min<size_t>(PTX_SCRATCH_MAX_BATCH, cap / npix), n_frames));
    for (uint32_t f0 = 0; f0 < n_frames; f0 += (uint32_t)batch) {
        const uint32_t nf = (uint32_t)std::min<size_t>(batch, n_frames - f0);
        TsBlock tb = {};
        std::memcpy(tb.v, timestamps + f0, nf * sizeof(int32_t));
        hipLaunchKernelGGL(k_store_timestamps, dim3(1), dim3(64), 0, ctx->stream, tb, d_ts, nf);
        HIP_TRY(ctx, hipGetLastError());
        if ((rc = render_batch(ctx, rb, nf, first_frame_number + f0, false))) return rc;   // untimed: creates no events
    }
    HIP_TRY(ctx, hipEventRecord(sl.ev_done, sl.stream));
    sl.busy = true;
    return PTX_OK;
}

extern "C" int ptx_synchronize(ptx_ctx* ctx) {
    if (!ctx) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_for_all(ctx, ptx_synchronize));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_all(ctx));
    drain_events(ctx);
    return PTX_OK;
}

extern "C" int ptx_frames_done(const ptx_ctx* ctx) {
    if (ctx && !ctx->subs.empty()) return multi_frames_done(ctx);
    return ctx ? (int)ctx->frames_done : PTX_ERR_INVALID;
}

static int read_accum(ptx_ctx* ctx, std::vector<float4>& h) {
    if (!ctx->have_params) return fail(ctx, PTX_ERR_STATE, "ptx_set_params first");
    int rc = ptx_synchronize(ctx);
    if (rc) return rc;
    h.resize((size_t)ctx->P.W * ctx->P.H);
    HIP_TRY(ctx, hipMemcpy(h.data(), ctx->d_accum, h.size() * sizeof(float4), hipMemcpyDeviceToHost));
    return PTX_OK;
}

extern "C" int ptx_read_sum_f32(ptx_ctx* ctx, float* rgb) {
    if (!ctx || !rgb) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_read(ctx, RD_SUM, rgb, 0, 0));
    std::vector<float4> h;
    int rc = read_accum(ctx, h);
    if (rc) return rc;
    for (size_t i = 0; i < h.size(); i++) { rgb[3 * i] = h[i].x; rgb[3 * i + 1] = h[i].y; rgb[3 * i + 2] = h[i].z; }
    return PTX_OK;
}

extern "C" int ptx_read_linear_f32(ptx_ctx* ctx, float* rgb) {
    if (!ctx || !rgb) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_read(ctx, RD_LINEAR, rgb, 0, 0));
    std::vector<float4> h;
    int rc = read_accum(ctx, h);
    if (rc) return rc;
    const float nf = (float)(ctx->frames_done ? ctx->frames_done : 1);
    for (size_t i = 0; i < h.size(); i++) { rgb[3 * i] = h[i].x / nf; rgb[3 * i + 1] = h[i].y / nf; rgb[3 * i + 2] = h[i].z / nf; }
    return PTX_OK;
}

extern "C" int ptx_read_rgba8(ptx_ctx* ctx, uint8_t* out, int apply_gamma, int flip_y) {
    if (!ctx || !out) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_read(ctx, RD_RGBA8, out, apply_gamma, flip_y));
    if (!ctx->have_params) return fail(ctx, PTX_ERR_STATE, "ptx_set_params first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)ctx->P.W * ctx->P.H;
    int rc = ensure_stage(ctx, npix * sizeof(uchar4));
    if (rc) return rc;
    uchar4* d_out = (uchar4*)ctx->d_stage;
    const float nf = (float)(ctx->frames_done ? ctx->frames_done : 1);
    HIP_TRY(ctx, sync_all(ctx));
    hipLaunchKernelGGL(k_pack_rgba8, dim3((uint32_t)((npix + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_accum, ctx->P.W, ctx->P.H, nf, apply_gamma, flip_y, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = sync_all(ctx);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, npix * sizeof(uchar4), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(ctx, PTX_ERR_HIP, std::string("ptx_read_rgba8: ") + hipGetErrorString(e));
    drain_events(ctx);
    return PTX_OK;
}

extern "C" int ptx_read_compat_rgba8(ptx_ctx* ctx, uint8_t* out, int apply_gamma, int flip_y) {
    if (!ctx || !out) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, fail(ctx, PTX_ERR_STATE, "ptx_read_compat_rgba8: the 8-bit progressive blend is sequential over frames; it needs a single-device context"));
    if (!ctx->have_params) return fail(ctx, PTX_ERR_STATE, "ptx_set_params first");
    if (!ctx->d_compat[0]) return fail(ctx, PTX_ERR_STATE, "ptx_read_compat_rgba8: no frame was rendered with option compat_rgba8 = 1");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)ctx->P.W * ctx->P.H;
    int rc = ensure_stage(ctx, npix * sizeof(uchar4));
    if (rc) return rc;
    uchar4* d_out = (uchar4*)ctx->d_stage;
    HIP_TRY(ctx, sync_all(ctx));
    hipLaunchKernelGGL(k_display_rgba8, dim3((uint32_t)((npix + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_compat[ctx->compat_cur], ctx->P.W, ctx->P.H,
                       apply_gamma, flip_y, ctx->opt_compat, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = sync_all(ctx);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, npix * sizeof(uchar4), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(ctx, PTX_ERR_HIP, std::string("ptx_read_compat_rgba8: ") + hipGetErrorString(e));
    drain_events(ctx);
    return PTX_OK;
}

// Checkpoint restore: the accumulation becomes `rgb` (W*H*3, as ptx_read_sum_f32 wrote it) with frames_done frames in it.
extern "C" int ptx_write_sum_f32(ptx_ctx* ctx, const float* rgb, uint32_t frames_done) {
    if (!ctx || !rgb) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_write_sum(ctx, rgb, frames_done));
    if (!ctx->have_params) return fail(ctx, PTX_ERR_STATE, "ptx_write_sum_f32: ptx_set_params first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t npix = (size_t)ctx->P.W * ctx->P.H;
    std::vector<float4> h(npix);
    for (size_t i = 0; i < npix; i++) h[i] = make_float4(rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2], 0.f);
    HIP_TRY(ctx, sync_all(ctx));
    HIP_TRY(ctx, hipMemcpy(ctx->d_accum, h.data(), npix * sizeof(float4), hipMemcpyHostToDevice));
    ctx->frames_done = frames_done;
    return PTX_OK;
}

extern "C" int ptx_trace_closest(ptx_ctx* ctx, const float* o, const float* d, uint32_t n, float* t, int32_t* tri_id) {
    if (!ctx) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, ptx_trace_closest(ctx->subs[0], o, d, n, t, tri_id));
    if (!ctx->have_scene) return fail(ctx, PTX_ERR_STATE, "ptx_trace_closest: ptx_upload_scene first");
    if (n == 0) return PTX_OK;
    if (!o || !d || !t || !tri_id) return fail(ctx, PTX_ERR_INVALID, "ptx_trace_closest: NULL array");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // staging: o | d | t | tri, each 256-B aligned inside the context's grow-only staging buffer
    const size_t sz3 = (((size_t)n * 12 + 255) / 256) * 256, sz1 = (((size_t)n * 4 + 255) / 256) * 256;
    int rc = ensure_stage(ctx, 2 * sz3 + 2 * sz1);
    if (rc) return rc;
    float* d_o = (float*)ctx->d_stage;
    float* d_d = (float*)((char*)ctx->d_stage + sz3);
    float* d_t = (float*)((char*)ctx->d_stage + 2 * sz3);
    int* d_tri = (int*)((char*)ctx->d_stage + 2 * sz3 + sz1);
    hipError_t e = hipMemcpy(d_o, o, (size_t)n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_d, d, (size_t)n * 12, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const uint32_t blocks = (n + 63) / 64;
        const bool pairs = ctx->opt_pipeline == 2 && ctx->pool_ok;      // the arithmetic of the renderer that is selected
#define PTX_TRACE(PR, PA) hipLaunchKernelGGL((k_trace<PR, PA>), dim3(blocks), dim3(64), 0, ctx->stream, scene_view(ctx), d_o, d_d, n, d_t, d_tri, ctx->d_counters)
        switch (prune_mode(ctx) * 2 + (pairs ? 1 : 0)) {
            case 0: PTX_TRACE(0, false); break;
            case 1: PTX_TRACE(0, true); break;
            case 2: PTX_TRACE(1, false); break;
            case 3: PTX_TRACE(1, true); break;
            case 4: PTX_TRACE(2, false); break;
            default: PTX_TRACE(2, true); break;
        }
#undef PTX_TRACE
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = sync_all(ctx);
    if (e == hipSuccess) e = hipMemcpy(t, d_t, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tri_id, d_tri, (size_t)n * 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(ctx, PTX_ERR_HIP, std::string("ptx_trace_closest: ") + hipGetErrorString(e));
    return PTX_OK;
}

extern "C" int ptx_selfcheck_pruning(ptx_ctx* ctx, uint32_t family, uint64_t n_rays, uint32_t seed, ptx_selfcheck_result* out) {
    if (!ctx || !out) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, ptx_selfcheck_pruning(ctx->subs[0], family, n_rays, seed, out));
    if (!ctx->have_scene) return fail(ctx, PTX_ERR_STATE, "ptx_selfcheck_pruning: ptx_upload_scene first");
    if (family > 11) return fail(ctx, PTX_ERR_INVALID, "ptx_selfcheck_pruning: family must be 0 .. 11");
    if (n_rays == 0 || n_rays > (1ull << 37)) return fail(ctx, PTX_ERR_INVALID, "ptx_selfcheck_pruning: n_rays must be in 1 .. 2^37");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_stage(ctx, sizeof(CheckOut));
    if (rc) return rc;
    CheckOut* d_out = (CheckOut*)ctx->d_stage;
    CheckOut h = {};
    for (int q = 0; q < 16; q++) h.first_tri[q][0] = h.first_tri[q][1] = -2;
    HIP_TRY(ctx, hipMemcpy(d_out, &h, sizeof h, hipMemcpyHostToDevice));
    CheckArgs A = {};
    A.n = n_rays; A.seed = seed; A.family = family; A.n_big = ctx->n_big_slots; A.big = ctx->d_big_slots;
    A.n_emit = ctx->n_emit_slots; A.emit = ctx->d_emit_slots;
    DScene V = scene_view(ctx);
    V.n_lbox = ctx->S.n_lbox;                                   // the light boxes are checked whatever the options would make of them
    A.bmin = {ctx->bbox_min[0], ctx->bbox_min[1], ctx->bbox_min[2]};
    A.bmax = {ctx->bbox_max[0], ctx->bbox_max[1], ctx->bbox_max[2]};
    const unsigned long long blocks = (n_rays + 63) / 64;
    const bool pairs = ctx->opt_pipeline == 2 && ctx->pool_ok;      // the arithmetic of the renderer that is selected
    // grid.x is 32 bits wide: launches of at most 2^30 blocks, each with its own seed offset
    unsigned long long done = 0;
    while (done < blocks) {
        const unsigned long long nb = std::min<unsigned long long>(blocks - done, 1ull << 30);
        CheckArgs B = A;
        B.n = std::min<unsigned long long>(n_rays - done * 64ull, nb * 64ull);
        B.seed = seed + (uint32_t)(done >> 30) * 0x632BE5ABu;
        const bool heuristic = ctx->opt_prune == 2;                 // the pruning rule that option "prune" selects
        if (pairs && heuristic) hipLaunchKernelGGL((k_selfcheck<2, true>), dim3((uint32_t)nb), dim3(64), 0, ctx->stream, V, B, d_out);
        else if (pairs) hipLaunchKernelGGL((k_selfcheck<1, true>), dim3((uint32_t)nb), dim3(64), 0, ctx->stream, V, B, d_out);
        else if (heuristic) hipLaunchKernelGGL((k_selfcheck<2, false>), dim3((uint32_t)nb), dim3(64), 0, ctx->stream, V, B, d_out);
        else hipLaunchKernelGGL((k_selfcheck<1, false>), dim3((uint32_t)nb), dim3(64), 0, ctx->stream, V, B, d_out);
        HIP_TRY(ctx, hipGetLastError());
        done += nb;
    }
    HIP_TRY(ctx, sync_all(ctx));
    HIP_TRY(ctx, hipMemcpy(&h, d_out, sizeof h, hipMemcpyDeviceToHost));
    out->rays = h.rays; out->hits = h.hits; out->mismatches = h.mismatches;
    out->emitter_hits = h.emitter_hits; out->anyhit_violations = h.anyhit_violations;
    std::memcpy(out->first, h.first, sizeof h.first);
    std::memcpy(out->first_tri, h.first_tri, sizeof h.first_tri);
    return PTX_OK;
}

extern "C" int ptx_primary_hits(ptx_ctx* ctx, int32_t* tri_id, float* t) {
    if (!ctx || !tri_id || !t) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, ptx_primary_hits(ctx->subs[0], tri_id, t));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_primary(ctx, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, sync_all(ctx));
    const size_t npix = (size_t)ctx->P.W * ctx->P.H;
    HIP_TRY(ctx, hipMemcpy(tri_id, ctx->d_prim_tri, npix * 4, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(t, ctx->d_prim_t, npix * 4, hipMemcpyDeviceToHost));
    return PTX_OK;
}

extern "C" int ptx_primary_dirs(ptx_ctx* ctx, float* dirs) {
    if (!ctx || !dirs) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, ptx_primary_dirs(ctx->subs[0], dirs));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->have_scene || !ctx->have_params) return fail(ctx, PTX_ERR_STATE, "scene and params must be set first");
    const size_t npix = (size_t)ctx->P.W * ctx->P.H;
    int rc = ensure_stage(ctx, npix * sizeof(float4));
    if (rc) return rc;
    float4* d_dirs = (float4*)ctx->d_stage;
    rc = ensure_primary(ctx, d_dirs);
    hipError_t e = sync_all(ctx);
    std::vector<float4> h(npix);
    if (rc == PTX_OK && e == hipSuccess) e = hipMemcpy(h.data(), d_dirs, npix * sizeof(float4), hipMemcpyDeviceToHost);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, PTX_ERR_HIP, std::string("ptx_primary_dirs: ") + hipGetErrorString(e));
    for (size_t i = 0; i < npix; i++) { dirs[3 * i] = h[i].x; dirs[3 * i + 1] = h[i].y; dirs[3 * i + 2] = h[i].z; }
    return PTX_OK;
}

extern "C" int ptx_get_counters(ptx_ctx* ctx, ptx_counters* out) {
    if (!ctx || !out) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_get_counters(ctx, out));
    int rc = ptx_synchronize(ctx);
    if (rc) return rc;
    unsigned long long h[CNT_NUM];
    HIP_TRY(ctx, hipMemcpy(h, ctx->d_counters, sizeof h, hipMemcpyDeviceToHost));
    out->samples = h[CNT_SAMPLES]; out->rays = h[CNT_RAYS]; out->rays_traced = h[CNT_RAYS_TRACED];
    out->node_visits = h[CNT_NODE_VISITS]; out->tri_tests = h[CNT_TRI_TESTS]; out->ray_hits = h[CNT_RAY_HITS];
    out->light_attempts = h[CNT_LIGHT_ATTEMPTS]; out->pixel_frames = h[CNT_PIXEL_FRAMES];
    out->kernel_launches = ctx->kernel_launches;
    out->kernel_ms = ctx->kernel_ms;
    for (int k = 0; k < 12; k++) out->diag[k] = h[CNT_WALK_USEFUL + k];
    return PTX_OK;
}

extern "C" int ptx_reset_counters(ptx_ctx* ctx) {
    if (!ctx) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_for_all(ctx, ptx_reset_counters));
    int rc = ptx_synchronize(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemset(ctx->d_counters, 0, CNT_NUM * sizeof(unsigned long long)));
    ctx->kernel_ms = 0.0;
    ctx->kernel_launches = 0;
    return PTX_OK;
}

extern "C" int ptx_accum_export_device(ptx_ctx* ctx, void* d_dst) {
    if (!ctx || !d_dst) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_accum_export(ctx, d_dst));
    if (!ctx->have_params) return fail(ctx, PTX_ERR_STATE, "ptx_set_params first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_all(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, ctx->d_accum, (size_t)ctx->P.W * ctx->P.H * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, sync_all(ctx));
    drain_events(ctx);
    return PTX_OK;
}

extern "C" int ptx_accum_import_device(ptx_ctx* ctx, const void* d_src, uint32_t frames_done) {
    if (!ctx || !d_src) return PTX_ERR_INVALID;
    PTX_MULTI(ctx, multi_accum_import(ctx, d_src, frames_done));
    if (!ctx->have_params) return fail(ctx, PTX_ERR_STATE, "ptx_set_params first");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, sync_all(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_accum, d_src, (size_t)ctx->P.W * ctx->P.H * sizeof(float4), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(ctx, sync_all(ctx));
    ctx->frames_done = frames_done;
    return PTX_OK;
}

extern "C" int ptx_get_size(const ptx_ctx* ctx, uint32_t* width, uint32_t* height) {
    if (!ctx || !width || !height) return PTX_ERR_INVALID;
    const ptx_ctx* c = ctx->subs.empty() ? ctx : ctx->subs[0];
    if (!c->have_params) return PTX_ERR_STATE;
    *width = c->P.W; *height = c->P.H;
    return PTX_OK;
}

// ------------------------------------------------------------------------------------------ multi-GPU
// Every (pixel, frame) is independent, so G GPUs render frames {g, g + G, ...} of the timestamp list over the full image
// into their own float accumulation buffers, and ONE all-reduce (RCCL over xGMI) of W*H*4 floats sums them.  Two shapes:
//   * one process, G devices:   ptx_create(device_ids, G) -- a routing context over G single-device contexts; the
//     communicators come from ncclCommInitAll, the all-reduce runs grouped on the subs' streams when a read-back needs the
//     sum (the Node host: the reference is one single-threaded JS process, main.js:391-438);
//   * one process per GPU:      ptx_comm_unique_id / ptx_comm_init / ptx_allreduce on ordinary single-device contexts
//     (bench.py under torch.distributed.run; the id travels over whatever channel the caller has).
// librccl is loaded on first use (dlopen), so single-GPU users never map it; an already mapped copy (torch bundles one)
// is preferred, so that the process keeps ONE RCCL and ONE HIP runtime.
#include <dlfcn.h>

namespace {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;       // NCCL_UNIQUE_ID_BYTES (rccl.h:40-43)
enum { NCCL_SUM = 0, NCCL_UINT64 = 5, NCCL_FLOAT32 = 7 };  // ncclRedOp_t / ncclDataType_t values of rccl.h:448-466
struct RcclApi {
    void* lib = nullptr;
    std::string path;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommCount)(const ncclComm_t, int*) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
std::mutex g_rccl_mutex;
RcclApi g_rccl;
std::string g_comm_error;

const RcclApi* rccl_api(std::string* err) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.lib) return &g_rccl;
    void* h = nullptr;
    std::string tried;
    const char* env = getenv("PTX_RCCL_LIB");
    if (env && *env) {       // an explicit choice is honoured or fails: no silent fall-through to another copy
        h = dlopen(env, RTLD_NOW | RTLD_GLOBAL); g_rccl.path = env;
        if (!h) { *err = std::string("cannot load librccl from PTX_RCCL_LIB (") + env + ": " + dlerror() + ")"; return nullptr; }
    }
    // a copy that is already mapped (torch's librccl.so has no SONAME, ROCm's is librccl.so.1)
    for (const char* name : {"librccl.so", "librccl.so.1"})
        if (!h && (h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) g_rccl.path = std::string(name) + " (already mapped)";
    for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"})
        if (!h) { h = dlopen(name, RTLD_NOW | RTLD_GLOBAL); g_rccl.path = name; if (!h) tried += std::string(name) + ": " + dlerror() + "; "; }
    if (!h) { *err = "cannot load librccl (" + tried + ")"; return nullptr; }
    RcclApi a;
    a.lib = h; a.path = g_rccl.path;
#define PTX_SYM(field, name) if (!(*(void**)(&a.field) = dlsym(h, name))) { *err = std::string("librccl lacks ") + name; return nullptr; }
    PTX_SYM(GetUniqueId, "ncclGetUniqueId") PTX_SYM(CommInitRank, "ncclCommInitRank") PTX_SYM(CommInitAll, "ncclCommInitAll")
    PTX_SYM(CommDestroy, "ncclCommDestroy") PTX_SYM(CommCount, "ncclCommCount") PTX_SYM(AllReduce, "ncclAllReduce")
    PTX_SYM(GroupStart, "ncclGroupStart") PTX_SYM(GroupEnd, "ncclGroupEnd") PTX_SYM(GetErrorString, "ncclGetErrorString")
#undef PTX_SYM
    g_rccl = a;
    return &g_rccl;
}
}  // namespace

#define NCCL_TRY(ctx, api, expr)                                                                         \
    do {                                                                                                 \
        const int r_ = (expr);                                                                           \
        if (r_ != 0) return fail((ctx), PTX_ERR_HIP, std::string(#expr) + ": " + (api)->GetErrorString(r_)); \
    } while (0)

// dst[p] = ((src[0][p] + src[1][p]) + src[2][p]) + ...  -- the all-reduce of contexts that share ONE device (fixed order)
struct SumSrcs { const float4* p[64]; int n; };
__global__ __launch_bounds__(256) void k_sum_accums(float4* __restrict__ dst, SumSrcs srcs, uint32_t npix) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= npix) return;
    float4 a = srcs.p[0][i];
    for (int g = 1; g < srcs.n; g++) {
        const float4 v = srcs.p[g][i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    dst[i] = a;
}

static void comm_release(ptx_ctx* ctx) {
    if (ctx->comm) {
        std::string err;
        const RcclApi* api = rccl_api(&err);
        if (api) (void)api->CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->comm_world = 0; ctx->comm_rank = -1;
    if (ctx->d_comm_cnt) { (void)hipFree(ctx->d_comm_cnt); ctx->d_comm_cnt = nullptr; }
}

// ---- one process per GPU
extern "C" int ptx_comm_unique_id(void* id128) {
    if (!id128) return PTX_ERR_INVALID;
    const RcclApi* api = rccl_api(&g_comm_error);
    if (!api) { g_create_error = "ptx_comm_unique_id: " + g_comm_error; return PTX_ERR_HIP; }
    ncclUniqueId id;
    const int r = api->GetUniqueId(&id);
    if (r != 0) { g_create_error = std::string("ncclGetUniqueId: ") + api->GetErrorString(r); return PTX_ERR_HIP; }
    std::memcpy(id128, id.internal, sizeof id.internal);
    return PTX_OK;
}

extern "C" int ptx_comm_init(ptx_ctx* ctx, const void* id128, int rank, int world) {
    if (!ctx) return PTX_ERR_INVALID;
    if (!ctx->subs.empty()) return fail(ctx, PTX_ERR_STATE, "ptx_comm_init: a multi-device context reduces by itself; communicators are for one context per process");
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(ctx, PTX_ERR_INVALID, "ptx_comm_init: need an id and 0 <= rank < world");
    std::string err;
    const RcclApi* api = rccl_api(&err);
    if (!api) return fail(ctx, PTX_ERR_HIP, "ptx_comm_init: " + err);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    comm_release(ctx);
    ncclUniqueId id;
    std::memcpy(id.internal, id128, sizeof id.internal);
    ncclComm_t comm = nullptr;
    NCCL_TRY(ctx, api, api->CommInitRank(&comm, world, id, rank));
    ctx->comm = comm; ctx->comm_world = world; ctx->comm_rank = rank;
    HIP_TRY(ctx, hipMalloc(&ctx->d_comm_cnt, sizeof(unsigned long long)));
    return PTX_OK;
}

extern "C" int ptx_comm_world(const ptx_ctx* ctx) {
    if (!ctx) return PTX_ERR_INVALID;
    if (!ctx->subs.empty()) return (int)ctx->subs.size();
    if (!ctx->comm) return 0;
    std::string err;
    const RcclApi* api = rccl_api(&err);
    int n = 0;
    if (!api || api->CommCount((ncclComm_t)ctx->comm, &n) != 0) return PTX_ERR_HIP;
    return n;                                           // what RCCL itself reports, not what the caller passed in
}

// Sums the accumulation buffers (W*H*4 f32) and the frame counts of all ranks with one grouped ncclAllReduce pair on the
// context's stream.  d_dst == NULL: in place -- afterwards every rank's accumulation IS the total and frames_done the total
// count (all-reduce once, at the end of the job).  d_dst != NULL: the total goes to that device buffer and the rank's own
// accumulation stays as it is (progress previews).  Blocks until done.
extern "C" int ptx_allreduce(ptx_ctx* ctx, void* d_dst, uint32_t* frames_total) {
    if (!ctx) return PTX_ERR_INVALID;
    if (!ctx->subs.empty()) return fail(ctx, PTX_ERR_STATE, "ptx_allreduce: a multi-device context reduces by itself when it is read");
    if (!ctx->have_params) return fail(ctx, PTX_ERR_STATE, "ptx_allreduce: ptx_set_params first");
    if (!ctx->comm) return fail(ctx, PTX_ERR_STATE, "ptx_allreduce: ptx_comm_init first");
    std::string err;
    const RcclApi* api = rccl_api(&err);
    if (!api) return fail(ctx, PTX_ERR_HIP, "ptx_allreduce: " + err);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t count = (size_t)ctx->P.W * ctx->P.H * 4;
    const unsigned long long mine = ctx->frames_done;
    HIP_TRY(ctx, sync_all(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_comm_cnt, &mine, sizeof mine, hipMemcpyHostToDevice, ctx->stream));
    NCCL_TRY(ctx, api, api->GroupStart());
    const int r1 = api->AllReduce(ctx->d_accum, d_dst ? d_dst : (void*)ctx->d_accum, count, NCCL_FLOAT32, NCCL_SUM, (ncclComm_t)ctx->comm, ctx->stream);
    const int r2 = api->AllReduce(ctx->d_comm_cnt, ctx->d_comm_cnt, 1, NCCL_UINT64, NCCL_SUM, (ncclComm_t)ctx->comm, ctx->stream);
    const int r3 = api->GroupEnd();
    if (r1 || r2 || r3) return fail(ctx, PTX_ERR_HIP, std::string("ncclAllReduce: ") + api->GetErrorString(r1 ? r1 : (r2 ? r2 : r3)));
    unsigned long long total = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&total, ctx->d_comm_cnt, sizeof total, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, sync_all(ctx));
    drain_events(ctx);
    if (!d_dst) ctx->frames_done = (uint32_t)total;
    if (frames_total) *frames_total = (uint32_t)total;
    return PTX_OK;
}

// ---- one process, G devices: the routing context
static int sub_fail(ptx_ctx* m, const ptx_ctx* sub, int rc) { m->err = sub->err; return rc; }

static ptx_ctx* multi_create(const int* device_ids, int n_devices) {
    bool all_same = true, all_distinct = true;
    for (int a = 0; a < n_devices; a++)
        for (int b = a + 1; b < n_devices; b++) {
            if (device_ids[a] == device_ids[b]) all_distinct = false; else all_same = false;
        }
    if (!all_same && !all_distinct) {
        g_create_error = "ptx_create: device_ids must be all distinct (RCCL all-reduce) or all the same device (one-GPU rehearsal)";
        return nullptr;
    }
    ptx_ctx* m = new ptx_ctx();
    m->device = device_ids[0];
    m->same_device = all_same;
    for (int g = 0; g < n_devices; g++) {
        ptx_ctx* sub = ptx_create(device_ids + g, 1);
        if (!sub) { multi_destroy(m); return nullptr; }          // g_create_error is set
        m->subs.push_back(sub);
    }
    return m;
}

static void multi_destroy(ptx_ctx* m) {
    if (!m->group_comms.empty()) {
        std::string err;
        const RcclApi* api = rccl_api(&err);
        for (void* c : m->group_comms) if (api && c) (void)api->CommDestroy((ncclComm_t)c);
    }
    std::vector<ptx_ctx*> subs;
    subs.swap(m->subs);                                       // ptx_destroy(sub) must not route
    for (ptx_ctx* s : subs) ptx_destroy(s);
    delete m;
}

static int multi_for_all(ptx_ctx* m, int (*fn)(ptx_ctx*)) {
    m->total_valid = false;
    for (ptx_ctx* s : m->subs) { const int rc = fn(s); if (rc) return sub_fail(m, s, rc); }
    return PTX_OK;
}
static int multi_upload_scene(ptx_ctx* m, const ptx_scene* sc) {
    for (ptx_ctx* s : m->subs) { const int rc = ptx_upload_scene(s, sc); if (rc) return sub_fail(m, s, rc); }
    return PTX_OK;
}
static int multi_set_params(ptx_ctx* m, const ptx_params* p) {
    m->total_valid = false;
    for (ptx_ctx* s : m->subs) { const int rc = ptx_set_params(s, p); if (rc) return sub_fail(m, s, rc); }
    return PTX_OK;
}
static int multi_set_option(ptx_ctx* m, const char* key, int64_t value) {
    if (key && std::string(key) == "compat_rgba8" && value != 0)
        return fail(m, PTX_ERR_INVALID, "ptx_set_option: compat_rgba8 blends frame after frame through one 8-bit target; it needs a single-device context");
    for (ptx_ctx* s : m->subs) { const int rc = ptx_set_option(s, key, value); if (rc) return sub_fail(m, s, rc); }
    return PTX_OK;
}

// frames {g, g + G, ...} of this call go to device g; every sub call only queues work, so the devices render concurrently
static int multi_render_frames(ptx_ctx* m, const int32_t* timestamps, uint32_t n_frames, uint32_t first_frame_number, bool noalloc) {
    if (!timestamps && n_frames) return fail(m, PTX_ERR_INVALID, "ptx_render_frames: timestamps is NULL");
    const uint32_t G = (uint32_t)m->subs.size();
    m->total_valid = false;
    std::vector<int32_t> mine;
    for (uint32_t g = 0; g < G; g++) {
        mine.clear();
        for (uint32_t f = g; f < n_frames; f += G) mine.push_back(timestamps[f]);
        if (mine.empty()) continue;
        ptx_ctx* s = m->subs[g];
        const int rc = noalloc ? ptx_render_frames_noalloc(s, mine.data(), (uint32_t)mine.size(), first_frame_number + g)
                               : ptx_render_frames(s, mine.data(), (uint32_t)mine.size(), first_frame_number + g);
        if (rc) return sub_fail(m, s, rc);
        // the subs of a one-GPU rehearsal share that GPU's static scratch: their no-alloc launches must not overlap
        if (noalloc && m->same_device) { const int rs = ptx_synchronize(s); if (rs) return sub_fail(m, s, rs); }
    }
    return PTX_OK;
}

static int multi_frames_done(const ptx_ctx* m) {
    int n = 0;
    for (const ptx_ctx* s : m->subs) n += (int)s->frames_done;
    return n;
}

// The sum of the subs' accumulations into subs[0]->d_total (every sub's d_total with RCCL): one grouped ncclAllReduce over
// the communicators of ncclCommInitAll, or k_sum_accums when the subs share one device.
static int multi_reduce(ptx_ctx* m) {
    if (m->total_valid) return PTX_OK;
    const size_t G = m->subs.size();
    ptx_ctx* s0 = m->subs[0];
    if (!s0->have_params) return fail(m, PTX_ERR_STATE, "ptx_set_params first");
    const size_t npix = (size_t)s0->P.W * s0->P.H;
    for (ptx_ctx* s : m->subs) {
        if (s->d_total) continue;
        HIP_TRY(m, hipSetDevice(s->device));
        HIP_TRY(m, hipMalloc(&s->d_total, npix * sizeof(float4)));
    }
    // Every sub renders on two alternating launch slots (streams), and the reduction below is queued on slot 0's stream
    // only: BOTH slots of EVERY sub (sub 0 included) must have finished their accumulates before d_accum is read.
    for (ptx_ctx* s : m->subs) {
        HIP_TRY(m, hipSetDevice(s->device));
        HIP_TRY(m, sync_all(s));
    }
    if (m->same_device) {
        SumSrcs srcs = {};
        srcs.n = (int)G;
        for (size_t g = 0; g < G; g++) srcs.p[g] = m->subs[g]->d_accum;
        HIP_TRY(m, hipSetDevice(s0->device));
        hipLaunchKernelGGL(k_sum_accums, dim3((uint32_t)((npix + 255) / 256)), dim3(256), 0, s0->stream, s0->d_total, srcs, (uint32_t)npix);
        HIP_TRY(m, hipGetLastError());
        HIP_TRY(m, sync_all(s0));
    } else {
        std::string err;
        const RcclApi* api = rccl_api(&err);
        if (!api) return fail(m, PTX_ERR_HIP, "multi-GPU reduction: " + err);
        if (m->group_comms.empty()) {
            std::vector<int> devs;
            for (ptx_ctx* s : m->subs) devs.push_back(s->device);
            std::vector<ncclComm_t> comms(G, nullptr);
            NCCL_TRY(m, api, api->CommInitAll(comms.data(), (int)G, devs.data()));
            for (ncclComm_t c : comms) m->group_comms.push_back(c);
        }
        NCCL_TRY(m, api, api->GroupStart());
        int bad = 0;
        for (size_t g = 0; g < G && !bad; g++) {
            ptx_ctx* s = m->subs[g];
            if (hipSetDevice(s->device) != hipSuccess) { bad = -1; break; }
            bad = api->AllReduce(s->d_accum, s->d_total, npix * 4, NCCL_FLOAT32, NCCL_SUM, (ncclComm_t)m->group_comms[g], s->stream);
        }
        const int end = api->GroupEnd();
        if (bad || end) return fail(m, PTX_ERR_HIP, std::string("ncclAllReduce (grouped): ") + (bad > 0 ? api->GetErrorString(bad) : (end ? api->GetErrorString(end) : "hipSetDevice failed")));
        for (ptx_ctx* s : m->subs) {
            HIP_TRY(m, hipSetDevice(s->device));
            HIP_TRY(m, sync_all(s));
        }
    }
    for (ptx_ctx* s : m->subs) drain_events(s);
    m->total_valid = true;
    return PTX_OK;
}

// read-backs see the total: subs[0] reads with its accumulation pointer and frame count swapped for the reduced ones
static int multi_read(ptx_ctx* m, int kind, void* out, int a, int b) {
    if (!out) return PTX_ERR_INVALID;
    int rc = multi_reduce(m);
    if (rc) return rc;
    ptx_ctx* s0 = m->subs[0];
    const uint32_t own_frames = s0->frames_done;
    std::swap(s0->d_accum, s0->d_total);
    s0->frames_done = (uint32_t)multi_frames_done(m);
    switch (kind) {
        case RD_LINEAR: rc = ptx_read_linear_f32(s0, (float*)out); break;
        case RD_SUM: rc = ptx_read_sum_f32(s0, (float*)out); break;
        default: rc = ptx_read_rgba8(s0, (uint8_t*)out, a, b); break;
    }
    std::swap(s0->d_accum, s0->d_total);
    s0->frames_done = own_frames;
    return rc ? sub_fail(m, s0, rc) : PTX_OK;
}

static int multi_write_sum(ptx_ctx* m, const float* rgb, uint32_t frames_done) {    // the checkpoint goes to device 0, the others start empty
    if (!rgb) return PTX_ERR_INVALID;
    m->total_valid = false;
    for (size_t g = 0; g < m->subs.size(); g++) {
        ptx_ctx* s = m->subs[g];
        const int rc = g == 0 ? ptx_write_sum_f32(s, rgb, frames_done) : ptx_reset_accum(s);
        if (rc) return sub_fail(m, s, rc);
    }
    return PTX_OK;
}

static int multi_get_counters(ptx_ctx* m, ptx_counters* out) {
    if (!out) return PTX_ERR_INVALID;
    std::memset(out, 0, sizeof *out);
    for (ptx_ctx* s : m->subs) {
        ptx_counters c;
        const int rc = ptx_get_counters(s, &c);
        if (rc) return sub_fail(m, s, rc);
        out->samples += c.samples; out->rays += c.rays; out->rays_traced += c.rays_traced; out->node_visits += c.node_visits;
        out->tri_tests += c.tri_tests; out->ray_hits += c.ray_hits; out->light_attempts += c.light_attempts; out->pixel_frames += c.pixel_frames;
        out->kernel_launches += c.kernel_launches;
        out->kernel_ms = std::max(out->kernel_ms, c.kernel_ms);       // the devices run side by side: the busiest one's time
        for (int k = 0; k < 12; k++) out->diag[k] += c.diag[k];
    }
    return PTX_OK;
}

static int multi_accum_export(ptx_ctx* m, void* d_dst) {               // the reduced total, from device subs[0]
    if (!d_dst) return PTX_ERR_INVALID;
    int rc = multi_reduce(m);
    if (rc) return rc;
    ptx_ctx* s0 = m->subs[0];
    HIP_TRY(m, hipSetDevice(s0->device));
    HIP_TRY(m, hipMemcpy(d_dst, s0->d_total, (size_t)s0->P.W * s0->P.H * sizeof(float4), hipMemcpyDeviceToDevice));
    return PTX_OK;
}
static int multi_accum_import(ptx_ctx* m, const void* d_src, uint32_t frames_done) {
    if (!d_src) return PTX_ERR_INVALID;
    m->total_valid = false;
    for (size_t g = 0; g < m->subs.size(); g++) {
        ptx_ctx* s = m->subs[g];
        const int rc = g == 0 ? ptx_accum_import_device(s, d_src, frames_done) : ptx_reset_accum(s);
        if (rc) return sub_fail(m, s, rc);
    }
    return PTX_OK;
}
