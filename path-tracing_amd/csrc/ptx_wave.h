// ptx_wave.h -- pipeline 1: the wave-scheduled ("wavefront inside the wave") path tracer (gfx950, wave64).
//
// Why: with one ray per lane traced to completion (pipeline 0) the 64 lanes of a wave wait for each other all the
// time -- rays through pixar_room's teapot visit ~400 nodes, the average ray 85 -- and rocprof shows the CU's
// vector-memory pipe (TA/TD) ~90 % busy issuing loads for waves with ~20 % of their lanes active.
//
// Here every lane is a small state machine with three kinds of pending work, kept in registers:
//     WALK  : one step of the threaded BVH walk (glsl:207-249: box test, hit -> i+1, miss -> missLink); a leaf whose
//             box is hit is appended to the lane's queue of pending leaves (PTX_WAVE_QUEUE entries)
//     TRI   : one Moeller-Trumbore test (glsl:150-173) of the next triangle of the lane's oldest pending leaf
//     SHADE : the ray is finished -> shade_step() (ptx_render.h) consumes the hit and produces the lane's next ray;
//             a lane whose (pixel, frame) task is finished writes its result and takes the next task of the wave
// Each iteration the wave VOTES (ballot + popcount) which phase to run, and only the lanes that have that kind of
// work execute it.  Lanes never wait for another lane's ray, leaf or path; they only wait for the vote.
// Order of triangle tests per ray is still increasing inorder slot, so strict '<' keeps the reference's tie rule.
// PRUNE uses the best t known when a node is visited (older or equal to the reference order's); PRUNE = 1 (provable,
// ptx_kernels.h) cuts inner nodes only here, PRUNE = 2 (heuristic margin) cuts leaves as well.
#pragma once
#include "ptx_render.h"

#ifndef PTX_WAVE_QUEUE
#define PTX_WAVE_QUEUE 4
#endif
#ifndef PTX_WAVE_SHADE_MIN
#define PTX_WAVE_SHADE_MIN 12     // run SHADE once this many lanes wait for it (or nothing else can run)
#endif
#ifndef PTX_WAVE_TRI_MIN
#define PTX_WAVE_TRI_MIN 48
#endif
#ifndef PTX_WAVE_MIN_WAVES
#define PTX_WAVE_MIN_WAVES 1      // __launch_bounds__ waves per SIMD (caps VGPRs: 4 -> 128, 5 -> 96)
#endif
#ifndef PTX_WAVE_WALK_STEPS
#define PTX_WAVE_WALK_STEPS 12    // node steps per WALK phase (a multiple of PTX_WAVE_WALK_UNROLL)
#endif
#ifndef PTX_WAVE_WALK_UNROLL
#define PTX_WAVE_WALK_UNROLL 2
#endif
#ifndef PTX_WAVE_WALK_KEEP
#define PTX_WAVE_WALK_KEEP 0      // leave the WALK phase early once fewer than KEEP/8 of the lanes it started with still walk (0: never)
#endif
#ifndef PTX_WAVE_TRI_KEEP
#define PTX_WAVE_TRI_KEEP 0
#endif
#ifndef PTX_WAVE_TRI_ROUNDS
#define PTX_WAVE_TRI_ROUNDS 6     // rounds of PTX_WAVE_LEAF_TRIS triangle tests per TRI phase
#endif
#ifndef PTX_WAVE_LEAF_TRIS
#define PTX_WAVE_LEAF_TRIS 2      // triangles of the current leaf tested per TRI phase
#endif

namespace ptx {

// ---- per-lane path state parked in LDS between SHADE phases (struct-of-arrays: word w of lane l at [w*64 + l],
//      conflict-free).  Only the SHADE phase touches it; WALK/TRI keep just the ray and the traversal cursor in
//      registers, which keeps the hot loop small (few loop-carried registers, higher occupancy).
// o and d are the lane's ray registers; dprim / prim_t / prim_tri are recomputed / reloaded from the pixel index.
enum { PW_SX = 0, PW_SY, PW_SUM, PW_S = PW_SUM + 3, PW_THR, PW_RAD = PW_THR + 3, PW_N = PW_RAD + 3, PW_COL = PW_N + 3, PW_LP = PW_COL + 3,
       PW_LPDF = PW_LP + 3, PW_LI, PW_PACK /* attempts | stage << 8 | bounce << 16 */, PW_PIX /* pixel | frame-in-batch << 26 */,
       PW_C_SAMPLES, PW_C_RAYS, PW_C_TRACED, PW_C_HITS, PW_C_ATT, PW_C_PF, PW_WORDS };

struct LdsLane {
    float* base;   // &lds[lane]
    __device__ __forceinline__ void st(int w, float v) const { base[w * 64] = v; }
    __device__ __forceinline__ void sti(int w, int v) const { base[w * 64] = __int_as_float(v); }
    __device__ __forceinline__ float ld(int w) const { return base[w * 64]; }
    __device__ __forceinline__ int ldi(int w) const { return __float_as_int(base[w * 64]); }
    __device__ __forceinline__ void st3(int w, V3 v) const { st(w, v.x); st(w + 1, v.y); st(w + 2, v.z); }
    __device__ __forceinline__ V3 ld3(int w) const { return mk(ld(w), ld(w + 1), ld(w + 2)); }
};

// Path accessors backed by the lane's LDS words (ptx_render.h: PTX_PATH_FIELDS)
struct PathLds {
    LdsLane L;
    V3& ro; V3& rd;                       // the ray lives in registers (the traversal needs it every step)
    const DParams& P;
    const float* prim_t_; const int* prim_tri_;
#define X3(f, W) __device__ __forceinline__ V3 f() const { return L.ld3(W); } __device__ __forceinline__ void set_##f(V3 v) const { L.st3(W, v); }
#define XF(f, W) __device__ __forceinline__ float f() const { return L.ld(W); } __device__ __forceinline__ void set_##f(float v) const { L.st(W, v); }
#define XI(f, W) __device__ __forceinline__ int f() const { return L.ldi(W); } __device__ __forceinline__ void set_##f(int v) const { L.sti(W, v); }
#define XU(f, W) __device__ __forceinline__ uint32_t f() const { return (uint32_t)L.ldi(W); } __device__ __forceinline__ void set_##f(uint32_t v) const { L.sti(W, (int)v); }
    XU(sx, PW_SX) XU(sy, PW_SY) X3(sum, PW_SUM) XI(s, PW_S) X3(thr, PW_THR) X3(rad, PW_RAD) X3(n, PW_N) X3(col, PW_COL) X3(lp, PW_LP)
    XF(lpdf, PW_LPDF) XI(li, PW_LI)
#undef X3
#undef XF
#undef XI
#undef XU
    __device__ __forceinline__ V3 o() const { return ro; }
    __device__ __forceinline__ void set_o(V3 v) const { ro = v; }
    __device__ __forceinline__ V3 d() const { return rd; }
    __device__ __forceinline__ void set_d(V3 v) const { rd = v; }
    __device__ __forceinline__ uint32_t pix() const { return (uint32_t)L.ldi(PW_PIX) & 0x03FFFFFFu; }
    __device__ __forceinline__ uint32_t frame() const { return (uint32_t)L.ldi(PW_PIX) >> 26; }
    __device__ __forceinline__ V3 dprim() const { const uint32_t px = pix(); return primary_dir(P, px % P.W, px / P.W); }
    __device__ __forceinline__ void set_dprim(V3) const {}
    __device__ __forceinline__ float prim_t() const { return prim_t_[pix()]; }
    __device__ __forceinline__ void set_prim_t(float) const {}
    __device__ __forceinline__ int prim_tri() const { return prim_tri_[pix()]; }
    __device__ __forceinline__ void set_prim_tri(int) const {}
    __device__ __forceinline__ int attempts() const { return L.ldi(PW_PACK) & 0xFF; }
    __device__ __forceinline__ void set_attempts(int v) const { L.sti(PW_PACK, (L.ldi(PW_PACK) & ~0xFF) | (v & 0xFF)); }
    __device__ __forceinline__ int stage() const { return (L.ldi(PW_PACK) >> 8) & 0xFF; }
    __device__ __forceinline__ void set_stage(int v) const { L.sti(PW_PACK, (L.ldi(PW_PACK) & ~0xFF00) | ((v & 0xFF) << 8)); }
    __device__ __forceinline__ int bounce() const { return (int)((uint32_t)L.ldi(PW_PACK) >> 16); }
    __device__ __forceinline__ void set_bounce(int v) const { L.sti(PW_PACK, (L.ldi(PW_PACK) & 0xFFFF) | (v << 16)); }
};

enum { F_ALIVE = 1, F_TASK = 2, F_TRAV = 4, F_CACHED = 8, F_INF = 16 /* 1/d has an infinite or NaN component */,
       F_PROBE = 32 /* pipeline 2: a light ray that has not yet tested the box of its blocker-cache leaf */ };

template <int PRUNE, bool MEMO, bool COUNT>
__global__ __launch_bounds__(64, PTX_WAVE_MIN_WAVES) void k_render_wave(DScene S, DParams P, const int32_t* __restrict__ timestamps,
                                                    const float* __restrict__ prim_t, const int* __restrict__ prim_tri,
                                                    float4* __restrict__ fb, uint32_t n_frames, uint32_t tasks_per_wave,
                                                    unsigned long long* counters) {
    __shared__ float lds[PW_WORDS * 64];
    const uint32_t lane = threadIdx.x;
    const LdsLane L{lds + lane};
    const uint32_t npix = P.W * P.H;
    const uint32_t tiles_x = (P.W + 7) >> 3, tiles_y = (P.H + 7) >> 3;
    const uint32_t tasks_per_frame = tiles_x * tiles_y * 64u;          // padded to whole 8x8 tiles
    const uint32_t total = tasks_per_frame * n_frames;                 // host guarantees < 2^32
    uint32_t next = blockIdx.x * tasks_per_wave;                       // wave-uniform cursor into the task list
    const uint32_t last = (total - next > tasks_per_wave) ? next + tasks_per_wave : total;

#pragma unroll
    for (int w = PW_C_SAMPLES; w <= PW_C_PF; w++) L.sti(w, 0);
    unsigned long long cyc[3] = {0, 0, 0}, its[3] = {0, 0, 0};
    uint32_t cnt_nodes = 0, cnt_tris = 0, d_walk = 0, d_tri = 0, d_shade = 0, d_walk_s = 0, d_tri_s = 0, d_shade_s = 0;   // COUNT only

    // registers carried through the loop: lane flags, the ray, the traversal cursor
    int flags = F_ALIVE;
    V3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 0.f), inv = mk(0.f, 0.f, 0.f);
    float best = 1.0e38f, limit = __builtin_inff();
    Kappa kappa = {0.f, 0.f};
    int best_tri = -1;
    uint32_t best_k = 0xFFFFFFFFu;
    uint32_t i = 0xFFFFFFFFu, nbase = 0;
    const uint32_t N = (uint32_t)S.N;
    int q[PTX_WAVE_QUEUE];
    int nq = 0;
    uint32_t k = 0, kend = 0;
#pragma unroll
    for (int e = 0; e < PTX_WAVE_QUEUE; e++) q[e] = 0;

    for (;;) {
        const bool trav = (flags & F_TRAV) != 0;
        const bool can_walk = trav && i < N && nq < PTX_WAVE_QUEUE;
        const bool has_tri = trav && (k < kend || nq > 0);
        const bool need_shade = (flags & F_ALIVE) && !(trav && (i < N || has_tri));   // ray finished, or no ray/task yet
        const int nW = __popcll(__ballot(can_walk)), nT = __popcll(__ballot(has_tri)), nS = __popcll(__ballot(need_shade));
        if ((nW | nT | nS) == 0) break;
        int phase;   // 0 walk, 1 tri, 2 shade
        if (nS >= PTX_WAVE_SHADE_MIN || (nW == 0 && nT == 0)) phase = 2;
        else if (nT >= PTX_WAVE_TRI_MIN || nW == 0) phase = 1;
        else if (nW >= nT || nT == 0) phase = 0;
        else phase = 1;
        phase = __builtin_amdgcn_readfirstlane(phase);
        unsigned long long t_ph = 0;
        if (COUNT) t_ph = __builtin_amdgcn_s_memtime();

        if (phase == 0) {
            if (COUNT) { d_walk += can_walk ? 1u : 0u; if (lane == 0) d_walk_s += 64u; }
            const int keepW = (nW * PTX_WAVE_WALK_KEEP) >> 3;
#pragma nounroll
            for (int g = 0; g < PTX_WAVE_WALK_STEPS / PTX_WAVE_WALK_UNROLL; g++) {
#pragma unroll
            for (int u = 0; u < PTX_WAVE_WALK_UNROLL; u++) {
                if (trav && i < N && nq < PTX_WAVE_QUEUE) {
                    const float4 a = ld16(S.nodes, nbase + (i << 5));
                    const float4 b = ld16(S.nodes, nbase + (i << 5) + 16);
                    if (COUNT) cnt_nodes++;
                    float tn;
                    const bool deg = (flags & F_INF) != 0;       // reference tree, SwiftShader's NaN rule
                    bool hit = deg ? slab_node_ref(ro, inv, a, b, tn) : slab_node(ro, inv, a, b, tn);
                    const int fc = __float_as_int(b.w);
                    if (PRUNE == 2) hit = hit && !(tn > limit);
                    if (PRUNE == 1) hit = hit && !(fc < 0 && prune_far<1>(tn, limit, (uint32_t)fc, kappa));
                    const bool push = hit && fc >= 0;
#pragma unroll
                    for (int e = 0; e < PTX_WAVE_QUEUE; e++) if (push && e == nq) q[e] = fc;
                    nq += push ? 1 : 0;
                    i = hit ? i + 1u : (uint32_t)__float_as_int(b.z);
                }
            }
            if (PTX_WAVE_WALK_KEEP && __popcll(__ballot(trav && i < N && nq < PTX_WAVE_QUEUE)) < keepW) break;
            }
        } else if (phase == 1) {
            if (COUNT) { d_tri += has_tri ? 1u : 0u; if (lane == 0) d_tri_s += 64u; }
            const int keepT = (nT * PTX_WAVE_TRI_KEEP) >> 3;
#pragma nounroll
            for (int r = 0; r < PTX_WAVE_TRI_ROUNDS; r++) {
            if (PTX_WAVE_TRI_KEEP && r && __popcll(__ballot(trav && (k < kend || nq > 0))) < keepT) break;
            if (trav && (k < kend || nq > 0)) {
                if (k == kend) {              // pop the oldest pending leaf
                    const int fc = q[0];
#pragma unroll
                    for (int e = 0; e + 1 < PTX_WAVE_QUEUE; e++) q[e] = q[e + 1];
                    nq--;
                    k = (uint32_t)(fc & 0x00FFFFFF);
                    kend = k + (uint32_t)((fc >> 24) & 0x7F);
                }
                // up to PTX_WAVE_LEAF_TRIS triangles of the leaf in one go: their loads are issued together (one memory
                // round trip per leaf instead of one per triangle); tests stay in increasing slot order
                float4 tv0[PTX_WAVE_LEAF_TRIS], te1[PTX_WAVE_LEAF_TRIS], te2[PTX_WAVE_LEAF_TRIS];
#pragma unroll
                for (int u = 0; u < PTX_WAVE_LEAF_TRIS; u++) {
                    if (k + u < kend) {
                        tv0[u] = ld16(S.tris, (k + u) * 48u);
                        te1[u] = ld16(S.tris, (k + u) * 48u + 16u);
                        te2[u] = ld16(S.tris, (k + u) * 48u + 32u);
                    }
                }
#pragma unroll
                for (int u = 0; u < PTX_WAVE_LEAF_TRIS; u++) {
                    if (k + u < kend) {
                        if (COUNT) cnt_tris++;
                        float t;
                        if (tri_test(ro, rd, mk(tv0[u]), mk(te1[u]), mk(te2[u]), t) && (t < best || (t == best && k + u < best_k))) {
                            best = t;
                            best_k = k + u;
                            best_tri = __float_as_int(tv0[u].w);
                            if (PRUNE) limit = prune_limit<PRUNE>(best, PRUNE == 2 ? ray_lat(S, inv) : 0.0f);
                        }
                    }
                }
                k = min(k + (uint32_t)PTX_WAVE_LEAF_TRIS, kend);
            }
            }
        } else {
            if (COUNT) { d_shade += need_shade ? 1u : 0u; if (lane == 0) d_shade_s += 64u; }
            bool run_shade = need_shade && (flags & F_TASK);      // a finished ray (or a pending cached primary hit)
            bool want_task = need_shade && !(flags & F_TASK);
            bool start_trav = false;
            LocalCnt c = {};
            PathLds p{L, ro, rd, P, prim_t, prim_tri};
            // (1) hand out the wave's next tasks.  Wave-uniform control flow: every lane executes the ballots, so the
            //     cursor `next` stays uniform.
            for (;;) {
                const unsigned long long m = __ballot(want_task);
                if (m == 0) break;
                const uint32_t mine = next + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                const uint32_t avail = last - next;                         // uniform; tasks left for this wave
                next += min((uint32_t)__popcll(m), avail);
                if (want_task) {
                    if (mine >= last) {
                        flags = 0; want_task = false;
                    } else {
                        const uint32_t frame = mine / tasks_per_frame;
                        const uint32_t r = mine - frame * tasks_per_frame;
                        uint32_t px, py;
                        if (tile_pixel(P, r >> 6, r & 63u, px, py)) {      // else: padding lane of an edge tile, draw again
                            const uint32_t pix = py * P.W + px;
                            L.sti(PW_PIX, (int)(pix | (frame << 26))); L.sti(PW_PACK, 0);
                            const int st = path_begin<MEMO>(P, p, px, py, timestamps[frame], 0.f, 0, c);
                            if (st == SH_DONE) {                            // max_path_length <= 0: black; draw again
                                fb[(size_t)frame * npix + pix] = make_float4(0.f, 0.f, 0.f, 1.0f);
                                L.sti(PW_C_PF, L.ldi(PW_C_PF) + 1);
                            } else {
                                flags |= F_TASK;
                                want_task = false;
                                if (st == SH_CACHED) { flags |= F_CACHED; run_shade = true; }
                                else { flags &= ~F_CACHED; start_trav = true; }
                            }
                        }
                    }
                }
            }
            // (2) consume finished rays / cached primary hits; a lane may go through several samples in a row when its
            //     primary ray misses or hits an emitter (every sample ends at once)
            if (run_shade) {
                bool hit = best < 1.0e38f;
                float t = best;
                int tri = best_tri;
                int st;
                bool cached = (flags & F_CACHED) != 0;
                for (;;) {
                    if (cached) { t = p.prim_t(); hit = t < 1.0e38f; tri = p.prim_tri(); }
                    st = shade_step<MEMO>(S, P, p, hit, t, tri, c);
                    cached = (st == SH_CACHED);
                    if (st != SH_CACHED) break;
                }
                flags &= ~F_CACHED;
                if (st == SH_DONE) {
                    const V3 rgb = path_result(P, p);
                    fb[(size_t)p.frame() * npix + p.pix()] = make_float4(rgb.x, rgb.y, rgb.z, 1.0f);
                    L.sti(PW_C_PF, L.ldi(PW_C_PF) + 1);
                    flags &= ~(F_TASK | F_TRAV);          // takes its next task in the next SHADE phase
                } else {
                    start_trav = true;
                }
            }
            // (3) start the traversal of the lanes that now hold a fresh ray
            if (start_trav) {
                inv = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                const bool deg = inv_degenerate(inv);
                flags = deg ? (flags | F_INF) : (flags & ~F_INF);
                nbase = deg ? S.ref_base : octant(rd) * S.oct_stride;
                best = 1.0e38f; limit = __builtin_inff(); best_tri = -1; best_k = 0xFFFFFFFFu;
                if (PRUNE == 1) kappa = ray_kappa(S, ro, rd);
                i = 0; nq = 0; k = 0; kend = 0;
                flags |= F_TRAV;
                c.rays_traced++;
            }
            if (need_shade) {
                L.sti(PW_C_SAMPLES, L.ldi(PW_C_SAMPLES) + (int)c.samples); L.sti(PW_C_RAYS, L.ldi(PW_C_RAYS) + (int)c.rays);
                L.sti(PW_C_TRACED, L.ldi(PW_C_TRACED) + (int)c.rays_traced); L.sti(PW_C_HITS, L.ldi(PW_C_HITS) + (int)c.ray_hits);
                L.sti(PW_C_ATT, L.ldi(PW_C_ATT) + (int)c.light_attempts);
            }
        }
        if (COUNT) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned long long dt = __builtin_amdgcn_s_memtime() - t_ph;
            if (lane == 0) { cyc[phase] += dt; its[phase]++; }
        }
    }
    if (COUNT && lane == 0) {
        for (int e = 0; e < 3; e++) { atomicAdd(&counters[CNT_CYC_WALK + e], cyc[e]); atomicAdd(&counters[CNT_IT_WALK + e], its[e]); }
    }
    LocalCnt c = {};
    c.samples = (uint32_t)L.ldi(PW_C_SAMPLES); c.rays = (uint32_t)L.ldi(PW_C_RAYS); c.rays_traced = (uint32_t)L.ldi(PW_C_TRACED);
    c.ray_hits = (uint32_t)L.ldi(PW_C_HITS); c.light_attempts = (uint32_t)L.ldi(PW_C_ATT);
    c.node_visits = cnt_nodes; c.tri_tests = cnt_tris;
    c.walk_useful = d_walk; c.walk_slots = d_walk_s; c.tri_useful = d_tri; c.tri_slots = d_tri_s; c.trace_useful = d_shade; c.trace_slots = d_shade_s;
    flush_counters(c, counters, (uint32_t)L.ldi(PW_C_PF));
}

} // namespace ptx
