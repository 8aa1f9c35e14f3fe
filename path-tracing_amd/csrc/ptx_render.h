// ptx_render.h -- the per-(pixel, frame) path-tracing state machine (gfx950).
//
// Restates main() + get_ray_radiance() + sample_random_light() + is_light_visible() of
// /root/reference/shaders/fragmentShaderPathTracing.glsl:353-529 as a stage handler around ONE kind of query:
// every ray the shader traces (path segment, light-visibility attempt, shadow ray) is a closest-hit query, so
// a path is a loop { closest hit of (o, d) ; shade_step() consumes the hit and produces the next ray }.
// The per-pixel seed chain (glsl:520-525) is kept by running a pixel-frame's samples sequentially in one lane.
//
// Exact memoisations (option "memo", template MEMO):
//  * the primary ray of a pixel is the same for every sample and frame (no jitter, glsl:514), so its hit
//    comes from the per-pixel primary-hit cache;
//  * the shadow ray of glsl:439 has the same origin and the same normalize(lightPoint - P) direction as
//    the last light attempt of glsl:385-386, so its closest hit is that attempt's closest hit.
// Both still count as rays of the reference algorithm (ptx_counters.rays); rays_traced counts traversals.
#pragma once
#include "ptx_kernels.h"

namespace ptx {

enum { ST_PATH = 0, ST_NEE = 1, ST_SHADOW = 2 };
enum { SH_TRACE = 0, SH_CACHED = 1, SH_DONE = 2 };   // what shade_step() wants next

// Path state behind accessors, so the same shade_step() runs on a register-resident path (pipeline 0: PathRegs)
// and on a path parked in LDS (pipeline 1: PathLds in ptx_wave.h, which keeps register pressure in the hot loop low).
#define PTX_PATH_FIELDS(X3, XF, XI, XU)                                                                          \
    XU(sx) XU(sy)            /* PCG2D state, chained through the task's samples */                               \
    X3(sum)                  /* sum of the samples' radiance (glsl:524)          */                               \
    X3(dprim) XF(prim_t) XI(prim_tri)   /* primary ray + its cached hit          */                               \
    XI(s)                    /* samples finished                                 */                               \
    X3(o) X3(d) X3(thr) X3(rad) X3(n) X3(col) X3(lp) XF(lpdf) XI(li) XI(attempts) XI(bounce) XI(stage)

struct PathRegs {
#define X3(f) V3 f##_; __device__ __forceinline__ V3 f() const { return f##_; } __device__ __forceinline__ void set_##f(V3 v) { f##_ = v; }
#define XF(f) float f##_; __device__ __forceinline__ float f() const { return f##_; } __device__ __forceinline__ void set_##f(float v) { f##_ = v; }
#define XI(f) int f##_; __device__ __forceinline__ int f() const { return f##_; } __device__ __forceinline__ void set_##f(int v) { f##_ = v; }
#define XU(f) uint32_t f##_; __device__ __forceinline__ uint32_t f() const { return f##_; } __device__ __forceinline__ void set_##f(uint32_t v) { f##_ = v; }
    PTX_PATH_FIELDS(X3, XF, XI, XU)
#undef X3
#undef XF
#undef XI
#undef XU
};
typedef PathRegs Path;

// two PCG2D draws through the accessors
template <class PT>
__device__ __forceinline__ void path_rng(PT& p, float& r0, float& r1) {
    uint32_t sx = p.sx(), sy = p.sy();
    pcg2d(sx, sy, r0, r1);
    p.set_sx(sx); p.set_sy(sy);
}

template <class PT>
__device__ __forceinline__ void path_new_sample(const DParams& P, PT& p) {
    p.set_thr(mk(1.f, 1.f, 1.f)); p.set_rad(mk(0.f, 0.f, 0.f));
    p.set_bounce(0); p.set_stage(ST_PATH); p.set_o(P.src); p.set_d(p.dprim());
}

// Starts the task of pixel (px, py) in the frame seeded by `timestamp`.  Returns SH_DONE when nothing is to be traced
// (max_path_length <= 0: glsl:408 never iterates, every sample is black), else SH_CACHED (MEMO) or SH_TRACE.
template <bool MEMO, class PT>
__device__ __forceinline__ int path_begin(const DParams& P, PT& p, uint32_t px, uint32_t py, int32_t timestamp, float prim_t, int prim_tri,
                                          LocalCnt& c) {
    const uint32_t k = ((uint32_t)timestamp) << 16;          // glsl:520
    p.set_sx(px ^ k); p.set_sy(py ^ k);
    p.set_sum(mk(0.f, 0.f, 0.f));
    p.set_dprim(primary_dir(P, px, py));
    p.set_prim_t(prim_t); p.set_prim_tri(prim_tri);
    p.set_s(0);
    p.set_n(mk(0.f, 0.f, 0.f)); p.set_col(mk(0.f, 0.f, 0.f)); p.set_lp(mk(0.f, 0.f, 0.f));
    p.set_lpdf(1.0f); p.set_li(0); p.set_attempts(0);
    path_new_sample(P, p);
    if (P.max_len <= 0) { c.samples += (uint32_t)P.spp; return SH_DONE; }
    c.samples++;
    return MEMO ? SH_CACHED : SH_TRACE;
}

// glsl:526-527
template <class PT>
__device__ __forceinline__ V3 path_result(const DParams& P, const PT& p) {
    const V3 m = p.sum() / (float)P.spp;
    return mk(clamp01(m.x), clamp01(m.y), clamp01(m.z));
}

// Consumes the closest hit (hit, t, tri) of the ray (p.o(), p.d()) in stage p.stage() and advances the path.
// Returns SH_TRACE: trace (p.o(), p.d()) next; SH_CACHED: the next hit is the pixel's cached primary hit (call again with
// it, no traversal); SH_DONE: all samples of the task are finished.
template <bool MEMO, class PT>
__device__ __forceinline__ int shade_step(const DScene& S, const DParams& P, PT& p, bool hit, float t, int tri, LocalCnt& c) {
    const int L = S.L;
    bool end_sample = false, do_bounce = false, do_direct = false, sh_hits = false;
    V3 sh_em = mk(0.f, 0.f, 0.f);

    if (p.stage() == ST_PATH) {
        c.rays++;
        if (!hit) {
            end_sample = true;                                            // glsl:487-488
        } else {
            c.ray_hits++;
            const V3 em = mk(S.attrE[tri]);
            p.set_col(mk(S.attrC[tri]));
            if (p.bounce() == 0 && (em.x > 0.0f || em.y > 0.0f || em.z > 0.0f)) {
                p.set_rad(p.col() * em);                                       // glsl:416-419
                end_sample = true;
            } else {
                p.set_n(mk(S.attrN[tri]));
                p.set_o(p.o() + p.d() * t);                                      // glsl:422 (P)
                if (L > 0) { p.set_attempts(0); p.set_stage(ST_NEE); }
                else do_bounce = true;                                    // L == 0: NEE skipped
            }
        }
    } else if (p.stage() == ST_NEE) {
        c.rays++;
        const V3 lem = mk(S.light[p.li() * 6 + 4]);
        V3 bem = mk(0.f, 0.f, 0.f);
        if (hit) { c.ray_hits++; if (tri >= 0) bem = mk(S.attrE[tri]); }   // tri < 0: a triangle known to have emission (0,0,0) (pipeline 2)
        const bool visible = !(hit && any_ne(bem, lem));                  // glsl:353-363
        if (visible || p.attempts() >= 10) {
            p.set_lpdf(visible ? S.light[p.li() * 6 + 5].x : 1.0f);            // glsl:389-395
            const V3 nL = mk(S.light[p.li() * 6 + 3]);
            // glsl:431: normalize(lightPoint - P) == the attempt's direction d (same expression)
            if (dot(p.n(), p.d()) > 0.0f && dot(nL, -p.d()) > 0.0f) {           // glsl:435
                if (MEMO) { c.rays++; if (hit) c.ray_hits++; sh_hits = hit; sh_em = bem; do_direct = true; do_bounce = true; }
                else p.set_stage(ST_SHADOW);                                 // trace glsl:439 for real
            } else {
                do_bounce = true;
            }
        }
        // else: another attempt (generated below)
    } else { // ST_SHADOW (only without MEMO)
        c.rays++;
        if (hit) { c.ray_hits++; if (tri >= 0) sh_em = mk(S.attrE[tri]); }
        sh_hits = hit;
        do_direct = true;
        do_bounce = true;
    }

    if (do_direct) {                                                      // glsl:442-453
        if (!sh_hits || (sh_em.x != 0.0f || sh_em.y != 0.0f || sh_em.z != 0.0f)) {
            const V3 nL = mk(S.light[p.li() * 6 + 3]);
            const V3 lem = mk(S.light[p.li() * 6 + 4]);
            const float dist = length(p.lp() - p.o());
            const float sa = fmaxf(dot(nL, -p.d()), 0.0f) / (dist * dist);
            const V3 brdf = (p.col() * PTX_M_1_PI) * dot(p.n(), p.d());
            const V3 direct = (lem * sa) * brdf;
            const float bpdf = p.lpdf() * sa;
            const float f1 = p.lpdf() * p.lpdf(), f2 = bpdf * bpdf;
            const float w = f1 / (f1 + f2);
            p.set_rad(p.rad() + ((p.thr() * w) * direct) / p.lpdf());
        }
    }

    if (do_bounce) {                                                      // glsl:459-486
        float r0, r1;
        path_rng(p, r0, r1);
        const V3 nd = sample_hemisphere(r0, r1, p.n());
        const float cos_theta = dot(nd, p.n());
        if (cos_theta < 0.0f) {
            end_sample = true;
        } else {
            const float pdf = cos_theta * PTX_M_1_PI;
            const V3 brdf = p.col() * PTX_M_1_PI;
            if (pdf > 0.0f) p.set_thr(p.thr() * ((brdf * cos_theta) / pdf));
            p.set_d(nd);
            bool rr_break = false;
            if (length(p.thr()) < 0.001f) {
                path_rng(p, r0, r1);
                if (r0 > 0.1f) rr_break = true;
                else p.set_thr(p.thr() / 0.1f);
            }
            p.set_bounce(p.bounce() + 1);
            if (rr_break || p.bounce() >= P.max_len) end_sample = true;
            else p.set_stage(ST_PATH);
        }
    }

    if (end_sample) {
        p.set_sum(p.sum() + p.rad());                                            // glsl:524
        p.set_s(p.s() + 1);
        if (p.s() == P.spp) return SH_DONE;
        c.samples++;
        path_new_sample(P, p);
        return MEMO ? SH_CACHED : SH_TRACE;
    }
    if (p.stage() == ST_NEE && !do_bounce) {
        // next light attempt (glsl:372-387); p.o() holds P
        p.set_attempts(p.attempts() + 1);
        c.light_attempts++;
        float r0, r1;
        path_rng(p, r0, r1);
        int ridx = (int)((float)L * r0);                                  // glsl:375: int(float(L)*r) % L
        if (ridx >= L) ridx -= L;                                         // only r == 1.0 reaches L
        p.set_li(ridx);
        path_rng(p, r0, r1);                                        // glsl:339-351
        if (r0 + r1 > 1.0f) { r0 = 1.0f - r0; r1 = 1.0f - r1; }
        const V3 v0 = mk(S.light[p.li() * 6 + 0]), v1 = mk(S.light[p.li() * 6 + 1]), v2 = mk(S.light[p.li() * 6 + 2]);
        p.set_lp((v0 * ((1.0f - r0) - r1) + v1 * r0) + v2 * r1);             // glsl:383
        p.set_d(normalize(p.lp() - p.o()));                                      // glsl:385
    }
    return SH_TRACE;
}

// One (pixel, frame) task, one lane, rays traced one after the other (pipeline 0).
template <int PRUNE, bool MEMO, bool COUNT>
__device__ __forceinline__ V3 render_pixel_frame(const DScene& S, const DParams& P, uint32_t px, uint32_t py, int32_t timestamp,
                                                 float prim_t, int prim_tri, LocalCnt& c) {
    Path p;
    int st = path_begin<MEMO>(P, p, px, py, timestamp, prim_t, prim_tri, c);
    while (st != SH_DONE) {
        float t = p.prim_t();
        int tri = p.prim_tri();
        bool hit = p.prim_t() < 1.0e38f;
        if (COUNT) { c.trace_useful += (st == SH_CACHED) ? 0u : 1u; if (first_active_lane()) c.trace_slots += 64u; }
        if (st != SH_CACHED) {
            hit = closest_hit<PRUNE, COUNT>(S, p.o(), p.d(), t, tri, c);
            c.rays_traced++;
        }
        st = shade_step<MEMO>(S, P, p, hit, t, tri, c);
    }
    return path_result(P, p);
}

} // namespace ptx
