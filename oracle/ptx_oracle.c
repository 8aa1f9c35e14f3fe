/*
 * ptx_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see ptx_oracle.h).
 *
 * Literal restatement of /root/reference/shaders/fragmentShaderPathTracing.glsl ("glsl:" below)
 * in strict binary32.  Build with:  gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp
 * (x86-64 SSE2 scalar float arithmetic is exact IEEE binary32; no x87, no FMA).
 *
 * Arithmetic contract (verified against the verbatim shader on SwiftShader):
 *   dot(a,b)   = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   cross(a,b) = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
 *   normalize(v) = v * (1.0f / sqrt(dot(v,v)));  length(v) = sqrt(dot(v,v))
 *   min(a,b)   = a < b ? a : b,  max(a,b) = a > b ? a : b  (a NaN operand yields b: SwiftShader's behaviour)
 *   float(uint) = two-step conversion of u32_to_f32 below (as SwiftShader does it)
 *   sin/cos    = the operation sequence of orc_sincos_2pi (GLSL leaves their precision open; it is the one SwiftShader
 *                evaluates, and the HIP kernels implement the same sequence so build-vs-oracle is bit-exact)
 */
#include "ptx_oracle.h"
#include <math.h>
#include <string.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scl(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 dvs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
    return V(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float length3(v3 a) { return sqrtf(dot(a, a)); }
static inline v3 normalize3(v3 a) { return scl(a, 1.0f / sqrtf(dot(a, a))); }
/* min/max with a NaN operand (GLSL: undefined) return the SECOND operand, as on the implementation the reference is run
 * on here (SwiftShader; probed with the derived debug shader: min(NaN,1) = 1, min(1,NaN) = NaN, max likewise).  This
 * decides which boxes a ray lying exactly in a bounding plane (0 * inf = NaN slabs) enters: glsl:183-187. */
static inline float fmin_(float a, float b) { return (a < b) ? a : b; }
static inline float fmax_(float a, float b) { return (a > b) ? a : b; }
static inline float clamp01(float x) { return fmin_(fmax_(x, 0.0f), 1.0f); }

#define M_PI_F 3.141592653589793238462643f     /* glsl:2 */
#define M_1_PI_F 0.3183098861837907f           /* glsl:3 */

/* ------------------------------------------------------------------ RNG (glsl:302-313, 520) */
/* vec2(uvec2): GLSL leaves the rounding of uint -> float open.  The contract follows the one implementation the
 * reference shader can be run on here (SwiftShader; pinned by tests/golden/ref_prim/ *_rad_b1 fixtures): words
 * below 2^31 are converted as signed integers (round to nearest even); words >= 2^31 as
 * float(int(x - 2^31)) + 2^31f, i.e. with TWO roundings.  The result differs from the single correctly rounded
 * conversion by one ulp for about 1 word in 16 of the upper half. */
static inline float u32_to_f32(uint32_t x) {
    float f = (float)(int32_t)(x & 0x7FFFFFFFu);
    return (x >> 31) ? f + 2147483648.0f : f;
}

void orc_pcg2d(uint32_t s[2], float out[2]) {
    s[0] = 1664525u * s[0] + 1013904223u;
    s[1] = 1664525u * s[1] + 1013904223u;
    s[0] += 1664525u * s[1];
    s[1] += 1664525u * s[0];
    s[0] ^= s[0] >> 16;
    s[1] ^= s[1] >> 16;
    s[0] += 1664525u * s[1];
    s[1] += 1664525u * s[0];
    s[0] ^= s[0] >> 16;
    s[1] ^= s[1] >> 16;
    out[0] = u32_to_f32(s[0]) * 2.32830643654e-10f;
    out[1] = u32_to_f32(s[1]) * 2.32830643654e-10f;
}

void orc_seed_init(uint32_t px, uint32_t py, int32_t timestamp, uint32_t seed[2]) {
    uint32_t k = ((uint32_t)timestamp) << 16; /* glsl:520 uvec2(timestamp << 16) */
    seed[0] = px ^ k;
    seed[1] = py ^ k;
}

/* ------------------------------------------------------------------ sincos contract
 * GLSL leaves the precision of sin/cos open; the contract is the sequence of separately rounded binary32 operations that
 * the implementation the reference is run on here (SwiftShader) evaluates, found by comparing float bits of 16384 angles
 * over [0, 2*pi] dumped by the derived debug shader (oracle/gen_ref_frames.py; fixture tests/golden/ref_prim/sincos.npz):
 * y = x / (2 pi) - round(..) in [-1/2, 1/2]; degree-7 sine and degree-6 cosine of the QUARTER angle pi*y/2; two angle
 * doublings, the second normalised by 1 / (s2^2 + c2^2); cos(x) = sin(x + pi/2).  |error| <= 7e-7. */
static inline float ss_sin(float x) {
    float y = x * 1.59154943e-1f;
    y = y - rintf(y);                                  /* round to nearest even */
    float y2 = y * y;
    float c1 = y2 * (y2 * (y2 * -0.0204391631f + 0.2536086171f) + -1.2336977925f) + 1.0f;
    float s1 = y * (y2 * (y2 * (y2 * -0.0046075748f + 0.0796819754f) + -0.645963615f) + 1.5707963235f);
    float c2 = c1 * c1 - s1 * s1;
    float s2 = (2.0f * s1) * c1;
    return ((2.0f * s2) * c2) * (1.0f / (s2 * s2 + c2 * c2));
}

void orc_sincos_2pi(float theta, float *so, float *co) {
    *so = ss_sin(theta);
    *co = ss_sin(theta + 1.57079632f);
}

/* ------------------------------------------------------------------ intersections */
static inline int ray_tri(v3 o, v3 d, v3 v0, v3 v1, v3 v2, float *out_t) { /* glsl:150-173 */
    v3 edge1 = sub(v1, v0);
    v3 edge2 = sub(v2, v0);
    v3 h = cross(d, edge2);
    float a = dot(edge1, h);
    if (a > -0.0001f && a < 0.0001f) return 0;
    float f = 1.0f / a;
    v3 s = sub(o, v0);
    float u = f * dot(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    v3 q = cross(s, edge1);
    float v = f * dot(d, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = f * dot(edge2, q);
    if (t > 0.0001f) { *out_t = t; return 1; }
    return 0;
}

static inline int ray_box(v3 o, v3 d, v3 bmin, v3 bmax) { /* glsl:175-191 */
    v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    v3 t1 = mul(sub(bmin, o), inv);
    v3 t2 = mul(sub(bmax, o), inv);
    v3 tmin = V(fmin_(t1.x, t2.x), fmin_(t1.y, t2.y), fmin_(t1.z, t2.z));
    v3 tmax = V(fmax_(t1.x, t2.x), fmax_(t1.y, t2.y), fmax_(t1.z, t2.z));
    float tNear = fmax_(fmax_(tmin.x, tmin.y), tmin.z);
    float tFar = fmin_(fmin_(tmax.x, tmax.y), tmax.z);
    return (tNear <= tFar) && (tFar > 0.0f);
}

int orc_ray_box(const float o[3], const float d[3], const float bmin[3], const float bmax[3]) {
    return ray_box(ld3(o), ld3(d), ld3(bmin), ld3(bmax));
}
int orc_ray_tri(const float o[3], const float d[3], const float v0[3], const float v1[3],
                const float v2[3], float *t) {
    return ray_tri(ld3(o), ld3(d), ld3(v0), ld3(v1), ld3(v2), t);
}

/* ---- EXPERIMENT (not the reference algorithm): ordered, pruned, stack-based traversal of the SAME tree, used only
 * to measure how much work an ordered traversal saves and whether its result equals the literal one.  Children of an
 * internal node i are L = i+1 and R = miss[L]; the winner is the lexicographic minimum of (t, inorder slot). */
static int g_traversal_mode = 0;   /* 0 literal, 1 literal + prune, 2 ordered + prune */
static float g_prune_rel = 0.001f, g_prune_abs = 0.001f;
void orc_set_traversal_mode(int mode, float rel, float abs_) { g_traversal_mode = mode; g_prune_rel = rel; g_prune_abs = abs_; }

static inline int ray_box_t(v3 o, v3 inv, v3 bmin, v3 bmax, float *tn) {
    v3 t1 = mul(sub(bmin, o), inv);
    v3 t2 = mul(sub(bmax, o), inv);
    float tNear = fmax_(fmax_(fmin_(t1.x, t2.x), fmin_(t1.y, t2.y)), fmin_(t1.z, t2.z));
    float tFar = fmin_(fmin_(fmax_(t1.x, t2.x), fmax_(t1.y, t2.y)), fmax_(t1.z, t2.z));
    *tn = tNear;
    return (tNear <= tFar) && (tFar > 0.0f);
}

static int bvh_closest_experimental(const orc_scene *s, v3 o, v3 d, float *out_t, int32_t *out_tri, orc_counters *c) {
    v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float best = 1.0e38f, limit = 3.0e38f;
    int32_t best_tri = -1, best_k = 0x7fffffff;
    const int32_t N = (int32_t)s->N;
    c->rays++;
    if (g_traversal_mode == 1) {
        int32_t i = 0;
        while (i != -1 && i < N) {
            float tn;
            c->node_visits++;
            if (ray_box_t(o, inv, ld3(s->node_min + 3 * i), ld3(s->node_max + 3 * i), &tn) && !(tn > limit)) {
                int32_t first = s->node_first[i];
                if (first != -2)
                    for (int32_t k = first; k < first + s->node_count[i]; k++) {
                        const float *p = s->coords + 9 * (size_t)s->inorder_tri[k];
                        float t;
                        c->tri_tests++;
                        if (ray_tri(o, d, ld3(p), ld3(p + 3), ld3(p + 6), &t) && t < best) { best = t; best_tri = s->inorder_tri[k]; limit = best * (1.0f + g_prune_rel) + g_prune_abs; }
                    }
                i++;
            } else i = s->node_miss[i];
        }
    } else {
        int32_t stack[64]; float stn[64]; int sp = 0;
        float tn;
        c->node_visits++;
        if (ray_box_t(o, inv, ld3(s->node_min), ld3(s->node_max), &tn)) { stack[sp] = 0; stn[sp++] = tn; }
        while (sp > 0) {
            int32_t i = stack[--sp];
            if (stn[sp] > limit) continue;
            int32_t first = s->node_first[i];
            if (first != -2) {
                for (int32_t k = first; k < first + s->node_count[i]; k++) {
                    const float *p = s->coords + 9 * (size_t)s->inorder_tri[k];
                    float t;
                    c->tri_tests++;
                    if (ray_tri(o, d, ld3(p), ld3(p + 3), ld3(p + 6), &t) && (t < best || (t == best && k < best_k))) {
                        best = t; best_k = k; best_tri = s->inorder_tri[k]; limit = best * (1.0f + g_prune_rel) + g_prune_abs;
                    }
                }
                continue;
            }
            int32_t L = i + 1, R = s->node_miss[L];
            float tl, tr;
            c->node_visits += 2;
            int hl = ray_box_t(o, inv, ld3(s->node_min + 3 * L), ld3(s->node_max + 3 * L), &tl) && !(tl > limit);
            int hr = ray_box_t(o, inv, ld3(s->node_min + 3 * R), ld3(s->node_max + 3 * R), &tr) && !(tr > limit);
            if (hl && hr) {
                if (tr < tl) { stack[sp] = L; stn[sp++] = tl; stack[sp] = R; stn[sp++] = tr; }
                else { stack[sp] = R; stn[sp++] = tr; stack[sp] = L; stn[sp++] = tl; }
            } else if (hl) { stack[sp] = L; stn[sp++] = tl; }
            else if (hr) { stack[sp] = R; stn[sp++] = tr; }
        }
    }
    *out_t = best; *out_tri = best_tri;
    if (best < 1.0e38f) { c->ray_hits++; return 1; }
    return 0;
}

/* diagnostic ray log of orc_trace_pixel (one thread): 8 words per traced ray -- o, d, t, tri (int bits) */
static __thread float *g_ray_log = 0;
static __thread uint32_t g_ray_log_cap = 0, g_ray_log_n = 0;
static int bvh_closest_literal(const orc_scene *s, v3 o, v3 d, float *out_t, int32_t *out_tri, orc_counters *c);

static int bvh_closest(const orc_scene *s, v3 o, v3 d, float *out_t, int32_t *out_tri, orc_counters *c) {
    int hit = g_traversal_mode ? bvh_closest_experimental(s, o, d, out_t, out_tri, c) : bvh_closest_literal(s, o, d, out_t, out_tri, c);
    if (g_ray_log) {
        if (g_ray_log_n < g_ray_log_cap) {
            float *r = g_ray_log + 8 * (size_t)g_ray_log_n;
            r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z; r[6] = *out_t;
            memcpy(r + 7, out_tri, 4);
        }
        g_ray_log_n++;
    }
    return hit;
}

/* glsl:207-249.  Returns hit flag; *tri = original triangle index of the closest hit. */
static int bvh_closest_literal(const orc_scene *s, v3 o, v3 d, float *out_t, int32_t *out_tri, orc_counters *c) {
    float best = 1.0e38f;
    int32_t best_tri = -1;
    int32_t i = 0;
    const int32_t N = (int32_t)s->N;
    c->rays++;
    while (i != -1 && i < N) {
        c->node_visits++;
        if (ray_box(o, d, ld3(s->node_min + 3 * i), ld3(s->node_max + 3 * i))) {
            int32_t first = s->node_first[i];
            if (first != -2) {
                int32_t cnt = s->node_count[i];
                for (int32_t k = first; k < first + cnt; k++) {
                    int32_t tri = s->inorder_tri[k];
                    const float *p = s->coords + 9 * (size_t)tri;
                    float t;
                    c->tri_tests++;
                    if (ray_tri(o, d, ld3(p), ld3(p + 3), ld3(p + 6), &t) && t < best) {
                        best = t;
                        best_tri = tri;
                    }
                }
            }
            i++;
        } else {
            i = s->node_miss[i];
        }
    }
    *out_t = best;
    *out_tri = best_tri;
    if (best < 1.0e38f) { c->ray_hits++; return 1; }
    return 0;
}

void orc_trace_closest(const orc_scene *s, const float *o, const float *d, uint32_t n, float *t_out,
                       int32_t *tri_out, orc_counters *c) {
    orc_counters local;
    memset(&local, 0, sizeof local);
    for (uint32_t i = 0; i < n; i++)
        bvh_closest(s, ld3(o + 3 * (size_t)i), ld3(d + 3 * (size_t)i), &t_out[i], &tri_out[i], &local);
    if (c) {
        c->rays += local.rays; c->node_visits += local.node_visits; c->tri_tests += local.tri_tests;
        c->ray_hits += local.ray_hits;
    }
}

/* ------------------------------------------------------------------ sampling */
static v3 sample_hemisphere(float r0, float r1, v3 n) { /* glsl:315-337 */
    float theta = (2.0f * M_PI_F) * r0;
    float r = sqrtf(r1);
    float sn, cs;
    orc_sincos_2pi(theta, &sn, &cs);
    float x = r * cs;
    float y = r * sn;
    float z = sqrtf(1.0f - r * r);
    v3 major = fabsf(n.x) < 0.57735026919f ? V(1, 0, 0) : (fabsf(n.y) < 0.57735026919f ? V(0, 1, 0) : V(0, 0, 1));
    v3 u = normalize3(cross(n, major));
    v3 v = cross(n, u);
    v3 dir = add(add(scl(u, x), scl(v, y)), scl(n, z));
    return normalize3(dir);
}

void orc_sample_hemisphere(const float rnd[2], const float n[3], float out[3]) {
    v3 r = sample_hemisphere(rnd[0], rnd[1], ld3(n));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

static inline int v3_ne(v3 a, v3 b) { return a.x != b.x || a.y != b.y || a.z != b.z; }

typedef struct { v3 v0, v1, v2, n, color, emission; } tri_t; /* glsl:11-18 */

static inline tri_t get_tri(const orc_scene *s, int32_t i) { /* glsl:91-115 */
    tri_t t;
    const float *p = s->coords + 9 * (size_t)i;
    t.v0 = ld3(p); t.v1 = ld3(p + 3); t.v2 = ld3(p + 6);
    t.n = ld3(s->normals + 3 * (size_t)i);
    t.color = ld3(s->colors + 3 * (size_t)i);
    t.emission = ld3(s->emissions + 3 * (size_t)i);
    return t;
}

/* glsl:404-491 */
static v3 ray_radiance(const orc_scene *s, const orc_params *p, v3 origin, v3 direction, uint32_t seed[2],
                       orc_counters *c) {
    v3 radiance = V(0, 0, 0);
    v3 thr = V(1, 1, 1);
    float rnd[2];
    const int32_t L = (int32_t)s->L;
    c->samples++;
    for (int i = 0; i < p->max_path_length; i++) {
        float t;
        int32_t tri_id;
        if (!bvh_closest(s, origin, direction, &t, &tri_id, c)) break;
        tri_t tri = get_tri(s, tri_id);
        if (i == 0 && (tri.emission.x > 0.0f || tri.emission.y > 0.0f || tri.emission.z > 0.0f))
            return mul(tri.color, tri.emission); /* glsl:416-419 */
        v3 P = add(origin, scl(direction, t));   /* glsl:422 */

        if (L > 0) { /* L == 0: '% 0' is undefined in GLSL; NEE is skipped (DESIGN.md "L = 0") */
            /* sample_random_light, glsl:365-396 */
            tri_t lt;
            v3 lp = V(0, 0, 0);
            float lpdf;
            int attempts = 0, visible = 0;
            float tb = 0.0f;
            int32_t blocker = -1;
            int hitsb = 0;
            memset(&lt, 0, sizeof lt);
            while (attempts < 10 && !visible) {
                attempts++;
                c->light_attempts++;
                orc_pcg2d(seed, rnd); c->rng_calls++;
                int32_t ridx = (int32_t)((float)L * rnd[0]) % L;          /* glsl:375 */
                lt = get_tri(s, s->light_tri[ridx]);
                orc_pcg2d(seed, rnd); c->rng_calls++;                       /* glsl:339-351 */
                float r1 = rnd[0], r2 = rnd[1];
                if (r1 + r2 > 1.0f) { r1 = 1.0f - r1; r2 = 1.0f - r2; }
                lp = add(add(scl(lt.v0, (1.0f - r1) - r2), scl(lt.v1, r1)), scl(lt.v2, r2)); /* glsl:383 */
                v3 dir = normalize3(sub(lp, P));
                /* is_light_visible, glsl:353-363 */
                hitsb = bvh_closest(s, P, dir, &tb, &blocker, c);
                visible = !(hitsb && v3_ne(ld3(s->emissions + 3 * (size_t)blocker), lt.emission));
            }
            if (visible) {
                float area = 0.5f * length3(cross(sub(lt.v1, lt.v0), sub(lt.v2, lt.v0)));
                lpdf = 1.0f / ((float)L * area);
            } else {
                lpdf = 1.0f;
            }
            /* glsl:431-455 */
            v3 toL = sub(lp, P);
            v3 Ldir = normalize3(toL);
            float dist = length3(toL);
            if (dot(tri.n, Ldir) > 0.0f && dot(lt.n, neg(Ldir)) > 0.0f) {
                float tb2; int32_t blk2;
                int hits = bvh_closest(s, P, Ldir, &tb2, &blk2, c); /* glsl:439 (same ray as the last attempt) */
                int lit = !hits;
                if (hits) {
                    v3 be = ld3(s->emissions + 3 * (size_t)blk2);
                    lit = (be.x != 0.0f || be.y != 0.0f || be.z != 0.0f);
                }
                if (lit) {
                    float sa = fmax_(dot(lt.n, neg(Ldir)), 0.0f) / (dist * dist);
                    v3 brdf = scl(scl(tri.color, M_1_PI_F), dot(tri.n, Ldir));
                    v3 direct = mul(scl(lt.emission, sa), brdf);
                    float bpdf = lpdf * sa;
                    float f1 = lpdf * lpdf, f2 = bpdf * bpdf;
                    float w = f1 / (f1 + f2);
                    radiance = add(radiance, dvs(mul(scl(thr, w), direct), lpdf));
                }
            }
        }

        orc_pcg2d(seed, rnd); c->rng_calls++;
        v3 nd = sample_hemisphere(rnd[0], rnd[1], tri.n); /* glsl:459 */
        float cos_theta = dot(nd, tri.n);
        if (cos_theta < 0.0f) break;
        float pdf = cos_theta * M_1_PI_F;
        v3 brdf = scl(tri.color, M_1_PI_F);
        if (pdf > 0.0f) thr = mul(thr, dvs(scl(brdf, cos_theta), pdf));
        origin = P;
        direction = nd;
        if (length3(thr) < 0.001f) { /* glsl:482-486 */
            orc_pcg2d(seed, rnd); c->rng_calls++;
            if (rnd[0] > 0.1f) break;
            thr = dvs(thr, 0.1f);
        }
    }
    return radiance;
}

void orc_primary_dir(const orc_params *p, uint32_t px, uint32_t py, float d[3]) { /* glsl:501-514,145-148 */
    float W = (float)p->width, H = (float)p->height;
    float aspect = (float)((double)p->width / (double)p->height); /* main.js:291 */
    float u = ((float)px + 0.5f) / W;
    float v = ((float)py + 0.5f) / H;
    if (aspect > 1.0f) { u = u * aspect; u = u - 0.125f; }
    else if (aspect < 1.0f) { v = v / aspect; v = v - 0.125f; }
    v3 ip = add(add(ld3(p->cam_left_bottom), scl(ld3(p->cam_right), u)), scl(ld3(p->cam_up), v));
    v3 dir = normalize3(sub(ip, ld3(p->cam_src)));
    d[0] = dir.x; d[1] = dir.y; d[2] = dir.z;
}

void orc_primary_hits(const orc_scene *s, const orc_params *p, float *t_out, int32_t *tri_out,
                      orc_counters *c) {
    orc_counters local;
    memset(&local, 0, sizeof local);
    v3 src = ld3(p->cam_src);
    for (uint32_t y = 0; y < p->height; y++)
        for (uint32_t x = 0; x < p->width; x++) {
            float d[3];
            size_t i = (size_t)y * p->width + x;
            orc_primary_dir(p, x, y, d);
            bvh_closest(s, src, ld3(d), &t_out[i], &tri_out[i], &local);
        }
    if (c) { c->rays += local.rays; c->node_visits += local.node_visits; c->tri_tests += local.tri_tests;
             c->ray_hits += local.ray_hits; }
}

static void acc_counters(orc_counters *dst, const orc_counters *src) {
    dst->samples += src->samples; dst->rays += src->rays; dst->node_visits += src->node_visits;
    dst->tri_tests += src->tri_tests; dst->ray_hits += src->ray_hits;
    dst->light_attempts += src->light_attempts; dst->rng_calls += src->rng_calls;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* one pixel of one frame (glsl:493-529), with every ray it traces written to ray_log (8 words each: o, d, t, tri);
 * returns the number of rays traced (may exceed cap: then only the first cap were stored).  Diagnostic. */
uint32_t orc_trace_pixel(const orc_scene *s, const orc_params *p, int32_t timestamp, uint32_t px, uint32_t py,
                         float rgb_out[3], float *ray_log, uint32_t cap) {
    float d[3];
    uint32_t seed[2];
    orc_counters local;
    memset(&local, 0, sizeof local);
    g_ray_log = ray_log; g_ray_log_cap = cap; g_ray_log_n = 0;
    orc_primary_dir(p, px, py, d);
    orc_seed_init(px, py, timestamp, seed);
    v3 sum = V(0, 0, 0);
    for (int sidx = 0; sidx != p->sample_count; ++sidx)
        sum = add(sum, ray_radiance(s, p, ld3(p->cam_src), ld3(d), seed, &local));
    sum = dvs(sum, (float)p->sample_count);
    rgb_out[0] = clamp01(sum.x); rgb_out[1] = clamp01(sum.y); rgb_out[2] = clamp01(sum.z);
    g_ray_log = 0;
    return g_ray_log_n;
}

void orc_render_frame(const orc_scene *s, const orc_params *p, int32_t timestamp, uint32_t y0,
                      uint32_t y1, float *rgb_out, orc_counters *c, int nthreads) {
    orc_render_rows(s, p, timestamp, y0, y1, 1, rgb_out, c, nthreads);
}

void orc_render_rows(const orc_scene *s, const orc_params *p, int32_t timestamp, uint32_t y0, uint32_t y1,
                     uint32_t y_step, float *rgb_out, orc_counters *c, int nthreads) { /* glsl:493-529 */
    if (y1 > p->height) y1 = p->height;
    if (y_step == 0) y_step = 1;
    orc_counters total;
    memset(&total, 0, sizeof total);
    v3 src = ld3(p->cam_src);
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
    {
        orc_counters local;
        memset(&local, 0, sizeof local);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int64_t y = (int64_t)y0; y < (int64_t)y1; y += (int64_t)y_step) {
            for (uint32_t x = 0; x < p->width; x++) {
                float d[3];
                uint32_t seed[2];
                orc_primary_dir(p, x, (uint32_t)y, d);
                orc_seed_init(x, (uint32_t)y, timestamp, seed);
                v3 sum = V(0, 0, 0);
                for (int sidx = 0; sidx != p->sample_count; ++sidx)
                    sum = add(sum, ray_radiance(s, p, src, ld3(d), seed, &local));
                sum = dvs(sum, (float)p->sample_count);
                float *o = rgb_out + 3 * ((size_t)y * p->width + x);
                o[0] = clamp01(sum.x); o[1] = clamp01(sum.y); o[2] = clamp01(sum.z);
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        acc_counters(&total, &local);
    }
    if (c) acc_counters(c, &total);
}

/* unorm8 store of a render target.  GL ES 3.0 2.1.6.1 allows either neighbour and prefers the nearest:
 * store 0 = round(c * 255) (the preferred rule; what GPUs do);
 * store 1 = the rule of the implementation the reference is run on here (SwiftShader; pinned byte for byte by
 *           tests/golden/ref_frames/ *_b1): a 16-bit intermediate c16 = trunc(c * 65535), then (c16 - (c16 >> 8) + 128) >> 8. */
static inline uint8_t to_unorm8(float c, int store) {
    if (store == 1) {
        uint32_t c16 = (uint32_t)(clamp01(c) * 65535.0f);
        return (uint8_t)((c16 - (c16 >> 8) + 0x80u) >> 8);
    }
    return (uint8_t)floorf(clamp01(c) * 255.0f + 0.5f);
}

/* pow(x, y) of the display pass (shaders/fragmentShaderOutput.glsl:35).  GLSL leaves the precision of pow open; the contract
 * is the operation sequence the implementation the reference is run on here evaluates, exp2(y * log2(x)) with the log2 and
 * exp2 below -- every float bit of 16384 arguments reproduced (tests/golden/ref_prim/probe_pow.npz), every separately
 * rounded.  log2: exponent and mantissa split, a (2,3) rational in the mantissa; exp2: floor split, degree-5 polynomial. */
static inline float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float ss_log2(float x) {
    const uint32_t xi = f_bits(x);
    float x1 = bits_f(((xi & 0x7F800000u) >> 8) | 0x3F800000u);
    x1 = (x1 - 1.4960938f) * 256.0f;                               /* the unbiased exponent */
    const float m = bits_f((xi & 0x007FFFFFu) | 0x3F800000u);       /* mantissa in [1, 2) */
    float x2 = (9.5428179e-2f * m + 4.7779095e-1f) * m + 1.9782813e-1f;
    const float x3 = ((1.6618466e-2f * m + 2.0350508e-1f) * m + 2.7382900e-1f) * m + 4.0496687e-2f;
    x2 = x2 / x3;
    return x1 + (m - 1.0f) * x2;
}
static inline float ss_exp2(float x) {
    float x0 = fmin_(x, bits_f(0x43010000u));                      /* 129 */
    x0 = fmax_(x0, bits_f(0xC2FDFFFFu));                            /* -126.99999 */
    const int32_t i = (int32_t)rintf(x0 - 0.5f);
    const float ii = bits_f((uint32_t)(i + 127) << 23);
    const float f = x0 - (float)i;
    float ff = bits_f(0x3AF61905u);
    ff = ff * f + bits_f(0x3C134806u);
    ff = ff * f + bits_f(0x3D64AA23u);
    ff = ff * f + bits_f(0x3E75EAD4u);
    ff = ff * f + bits_f(0x3F31727Bu);
    ff = ff * f + 1.0f;
    return ii * ff;
}
float orc_pow(float x, float y) { return ss_exp2(y * ss_log2(x)); }

/* The display pass over an 8-bit target (fragmentShaderOutput.glsl:22-41 drawn into the canvas, main.js:357-373):
 * c = texel / 255, pow(c, 1 / 2.2), stored with the canvas' float -> unorm8 rule (store as in to_unorm8). */
void orc_display_rgba8(const uint8_t *in_rgba8, uint32_t npix, int32_t store, uint8_t *out_rgba8) {
    for (uint32_t i = 0; i < npix; i++) {
        for (int k = 0; k < 3; k++)
            out_rgba8[4 * (size_t)i + k] = to_unorm8(orc_pow((float)in_rgba8[4 * (size_t)i + k] / 255.0f, 1.0f / 2.2f), store);
        out_rgba8[4 * (size_t)i + 3] = 255;
    }
}

void orc_blend_rgba8(const orc_params *p, const float *cur_rgb, const uint8_t *prev_rgba8,
                     int32_t frame_number, uint8_t *out_rgba8) {
    orc_blend_rgba8_store(p, cur_rgb, prev_rgba8, frame_number, 0, out_rgba8);
}

void orc_blend_rgba8_store(const orc_params *p, const float *cur_rgb, const uint8_t *prev_rgba8,
                           int32_t frame_number, int32_t store, uint8_t *out_rgba8) { /* glsl:501-512,532-552 */
    const uint32_t W = p->width, H = p->height;
    float aspect = (float)((double)W / (double)H);
    float bf = 1.0f / (float)(frame_number + 1);
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) {
            size_t i = (size_t)y * W + x;
            float prev[3] = {0, 0, 0};
            if (prev_rgba8) {
                float u = ((float)x + 0.5f) / (float)W, v = ((float)y + 0.5f) / (float)H;
                if (aspect > 1.0f) { u = u * aspect; u = u - 0.125f; }
                else if (aspect < 1.0f) { v = v / aspect; v = v - 0.125f; }
                /* NEAREST + REPEAT (WebGL default wrap; modules/bufferManager.js:19-20 sets only filters) */
                float fu = u - floorf(u), fv = v - floorf(v);
                int32_t sx = (int32_t)floorf(fu * (float)W), sy = (int32_t)floorf(fv * (float)H);
                if (store == 1) { /* SwiftShader's NEAREST addressing: 16-bit truncated coordinate, texel = (c16 * size) >> 16;
                                     differs from the line above only where the coordinate sits on a texel boundary */
                    uint32_t u16 = (uint32_t)(fu * 65536.0f), v16 = (uint32_t)(fv * 65536.0f);
                    if (u16 > 65535u) u16 = 65535u;
                    if (v16 > 65535u) v16 = 65535u;
                    sx = (int32_t)(((uint64_t)u16 * W) >> 16); sy = (int32_t)(((uint64_t)v16 * H) >> 16);
                }
                if (sx >= (int32_t)W) sx = (int32_t)W - 1;
                if (sy >= (int32_t)H) sy = (int32_t)H - 1;
                const uint8_t *q = prev_rgba8 + 4 * ((size_t)sy * W + sx);
                for (int k = 0; k < 3; k++) prev[k] = clamp01((float)q[k] / 255.0f);
            }
            for (int k = 0; k < 3; k++) {
                float cur = cur_rgb[3 * i + k];
                float b = prev[k] * (1.0f - bf) + cur * bf; /* mix(x,y,a) = x*(1-a) + y*a */
                out_rgba8[4 * i + k] = to_unorm8(b, store);
            }
            out_rgba8[4 * i + 3] = 255;
        }
}
