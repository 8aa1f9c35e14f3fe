// ptx_render.h -- the per-(pixel, frame) path-tracing state machine (gfx950).
//
// Restates main() + get_ray_radiance() + sample_random_light() + is_light_visible() of
// /root/reference/shaders/fragmentShaderPathTracing.glsl:353-529 as ONE loop around ONE closest-hit
// traversal site.  Every ray the shader traces (path segment, light-visibility attempt, shadow ray) is a
// closest-hit query, so a lane is always either traversing or running a short stage handler that
// produces its next ray; lanes of a wave in different stages still share the traversal loop.
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

// One (pixel, frame) task: returns clamp(mean of spp path radiances, 0, 1)  (glsl:516-529).
template <bool PRUNE, bool MEMO, bool COUNT>
__device__ __forceinline__ V3 render_pixel_frame(const DScene& S, const DParams& P, uint32_t px, uint32_t py, int32_t timestamp,
                                                 float prim_t, int prim_tri, LocalCnt& c) {
    const V3 dprim = primary_dir(P, px, py);
    const uint32_t k = ((uint32_t)timestamp) << 16;          // glsl:520
    uint32_t sx = px ^ k, sy = py ^ k;
    V3 sum = mk(0.f, 0.f, 0.f);
    const int L = S.L;
    const float fL = (float)L;

    // per-path state
    V3 o = P.src, d = dprim;
    V3 thr = mk(1.f, 1.f, 1.f), rad = mk(0.f, 0.f, 0.f);
    V3 n = mk(0.f, 0.f, 0.f), col = mk(0.f, 0.f, 0.f);        // attributes of the current path vertex
    V3 lp = mk(0.f, 0.f, 0.f);                                 // light point of the current attempt
    float lpdf = 1.0f;
    int li = 0, attempts = 0;
    int bounce = 0, s = 0, stage = ST_PATH;
    bool alive = (P.spp > 0);
    bool cached = MEMO;                                        // next PATH result comes from the primary cache
    if (P.max_len <= 0) {                                      // glsl:408 never iterates: every sample is black
        c.samples += (uint32_t)(P.spp > 0 ? P.spp : 0);
        alive = false;
    }
    if (alive) c.samples++;

    while (alive) {
        float t = prim_t;
        int tri = prim_tri;
        bool hit = prim_t < 1.0e38f;
        if (!(MEMO && cached)) {
            hit = closest_hit<PRUNE, COUNT>(S, o, d, t, tri, c);
            c.rays_traced++;
        }
        cached = false;

        bool end_sample = false, do_bounce = false;
        bool sh_hits = false;
        V3 sh_em = mk(0.f, 0.f, 0.f);
        bool do_direct = false;

        if (stage == ST_PATH) {
            c.rays++;
            if (!hit) {
                end_sample = true;                                            // glsl:487-488
            } else {
                c.ray_hits++;
                const V3 em = mk(S.attrE[tri]);
                col = mk(S.attrC[tri]);
                if (bounce == 0 && (em.x > 0.0f || em.y > 0.0f || em.z > 0.0f)) {
                    rad = col * em;                                           // glsl:416-419
                    end_sample = true;
                } else {
                    n = mk(S.attrN[tri]);
                    o = o + d * t;                                            // glsl:422 (P)
                    if (L > 0) { attempts = 0; stage = ST_NEE; }
                    else do_bounce = true;                                    // L == 0: NEE skipped
                }
            }
        } else if (stage == ST_NEE) {
            c.rays++;
            const V3 lem = mk(S.light[li * 6 + 4]);
            V3 bem = mk(0.f, 0.f, 0.f);
            if (hit) { c.ray_hits++; bem = mk(S.attrE[tri]); }
            const bool visible = !(hit && any_ne(bem, lem));                  // glsl:353-363
            if (visible || attempts >= 10) {
                lpdf = visible ? S.light[li * 6 + 5].x : 1.0f;                // glsl:389-395
                const V3 nL = mk(S.light[li * 6 + 3]);
                // glsl:431: normalize(lightPoint - P) == the attempt's direction d (same expression)
                if (dot(n, d) > 0.0f && dot(nL, -d) > 0.0f) {                 // glsl:435
                    if (MEMO) { c.rays++; if (hit) c.ray_hits++; sh_hits = hit; sh_em = bem; do_direct = true; do_bounce = true; }
                    else stage = ST_SHADOW;                                   // trace glsl:439 for real
                } else {
                    do_bounce = true;
                }
            }
            // else: another attempt (generated below)
        } else { // ST_SHADOW (only without MEMO)
            c.rays++;
            if (hit) { c.ray_hits++; sh_em = mk(S.attrE[tri]); }
            sh_hits = hit;
            do_direct = true;
            do_bounce = true;
        }

        if (do_direct) {                                                      // glsl:442-453
            if (!sh_hits || (sh_em.x != 0.0f || sh_em.y != 0.0f || sh_em.z != 0.0f)) {
                const V3 nL = mk(S.light[li * 6 + 3]);
                const V3 lem = mk(S.light[li * 6 + 4]);
                const float dist = length(lp - o);
                const float sa = fmaxf(dot(nL, -d), 0.0f) / (dist * dist);
                const V3 brdf = (col * PTX_M_1_PI) * dot(n, d);
                const V3 direct = (lem * sa) * brdf;
                const float bpdf = lpdf * sa;
                const float f1 = lpdf * lpdf, f2 = bpdf * bpdf;
                const float w = f1 / (f1 + f2);
                rad = rad + ((thr * w) * direct) / lpdf;
            }
        }

        if (do_bounce) {                                                      // glsl:459-486
            float r0, r1;
            pcg2d(sx, sy, r0, r1);
            const V3 nd = sample_hemisphere(r0, r1, n);
            const float cos_theta = dot(nd, n);
            if (cos_theta < 0.0f) {
                end_sample = true;
            } else {
                const float pdf = cos_theta * PTX_M_1_PI;
                const V3 brdf = col * PTX_M_1_PI;
                if (pdf > 0.0f) thr = thr * ((brdf * cos_theta) / pdf);
                d = nd;
                bool rr_break = false;
                if (length(thr) < 0.001f) {
                    pcg2d(sx, sy, r0, r1);
                    if (r0 > 0.1f) rr_break = true;
                    else thr = thr / 0.1f;
                }
                bounce++;
                if (rr_break || bounce >= P.max_len) end_sample = true;
                else stage = ST_PATH;
            }
        }

        if (end_sample) {
            sum = sum + rad;                                                  // glsl:524
            s++;
            if (s == P.spp) {
                alive = false;
            } else {
                c.samples++;
                thr = mk(1.f, 1.f, 1.f); rad = mk(0.f, 0.f, 0.f);
                bounce = 0; stage = ST_PATH; o = P.src; d = dprim;
                cached = MEMO;
            }
        } else if (stage == ST_NEE && !do_bounce) {
            // next light attempt (glsl:372-387); o holds P
            attempts++;
            c.light_attempts++;
            float r0, r1;
            pcg2d(sx, sy, r0, r1);
            int ridx = (int)(fL * r0);                                        // glsl:375: int(float(L)*r) % L
            if (ridx >= L) ridx -= L;                                         // only r == 1.0 reaches L
            li = ridx;
            pcg2d(sx, sy, r0, r1);                                            // glsl:339-351
            if (r0 + r1 > 1.0f) { r0 = 1.0f - r0; r1 = 1.0f - r1; }
            const V3 v0 = mk(S.light[li * 6 + 0]), v1 = mk(S.light[li * 6 + 1]), v2 = mk(S.light[li * 6 + 2]);
            lp = (v0 * ((1.0f - r0) - r1) + v1 * r0) + v2 * r1;               // glsl:383
            d = normalize(lp - o);                                            // glsl:385
        }
    }

    sum = sum / (float)P.spp;                                                 // glsl:526
    return mk(clamp01(sum.x), clamp01(sum.y), clamp01(sum.z));                // glsl:527
}

} // namespace ptx
