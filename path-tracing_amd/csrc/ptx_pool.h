// ptx_pool.h -- pipeline 2: wave-scheduled path tracer with a SHARED leaf queue (gfx950, wave64).
//
// Same idea as ptx_wave.h (every lane a persistent state machine, the wave votes which kind of work runs next), but the
// triangle work is decoupled from the lane that found it.  A lane whose walk hits a leaf box appends an item
// {first slot | owner lane, slab entry distance} to a queue in LDS that belongs to the whole wave.  The TRI phase hands
// the 64 oldest items to the 64 lanes, whoever owns them: a lane fetches the owner's ray out of the owner's registers
// (ds_bpermute), tests the leaf's triangles (glsl:150-173) and merges its best hit into the owner's 64-bit key
// (t bits << 32 | inorder slot) with one LDS min -- the lexicographic (t, slot) minimum the reference's strict '<' in
// increasing slot order produces.  So TRI phases run with every lane busy whenever the wave has 64 leaves pending, no
// matter how unevenly the rays produced them, and the walk never stalls on a full per-lane queue.
// With PRUNE an item is dropped when its leaf is out of reach of the owner's best hit by the time it is dequeued
// (prune_far, ptx_kernels.h: provably for PRUNE = 1, by the heuristic margin for PRUNE = 2; front-to-back order makes this
// frequent).  Inner nodes are cut in the WALK phase by the same rule with the coefficient stored in the node.
// A ray is finished when its walk has ended and the queue head has passed its newest item (items are consumed in order).
// Light-visibility rays end early (S.n_lbox > 0): a TRI lane that finds a hit on a triangle without emission, nearer than any
// emissive triangle can be (light_bound, ptx_kernels.h), stores it with distance 0; that drops the ray's queued leaves at
// dequeue and ends its walk at the next WALK phase.  The key carries the emission flag of its triangle in bit 0
// (t bits << 32 | slot << 1 | flag: still ordered by (t, slot)).
// Blocker cache: 92 % of the light attempts are blocked and the next attempt of the same bounce (same origin, a direction a few
// degrees off) is blocked in the same leaf in 59 % of the cases, in the lane's last different blocker leaf before that in 24 %.
// A lane remembers the leaves of its last two different blockers (reference-tree node indices, looked up by slot in the
// triangle record; one entry for trees under 256 or over 65 534 nodes); when its next attempt starts, both leaves' boxes are
// tested at once and every one that is hit goes into the queue before the walk has taken a step (the reference tests such a
// leaf too: its box passes, hence every ancestor's).  A conclusive hit in the next TRI phase ends the ray after at most one
// WALK phase (forcing a TRI phase right after such a SHADE phase was measured: no gain).
#pragma once
#include "ptx_wave.h"
#include <type_traits>

#ifndef PTX_POOL_QCAP
#define PTX_POOL_QCAP 128        // queue entries per wave (power of two, >= 128: one WALK step may add 64)
#endif
#ifndef PTX_POOL_WALK_STEPS
#define PTX_POOL_WALK_STEPS 16   // node steps per WALK phase (ends early when the queue could overflow)
#endif
#ifndef PTX_POOL_SHADE_W
#define PTX_POOL_SHADE_W 4       // a lane waiting for SHADE weighs this many QUARTERS of a walking lane / queued leaf in the vote (round 3, with the
                                 // two-entry blocker cache: 2 / 3 / 4 / 5 / 6 / 8 -> 59.3 / 59.1 / 58.4 / 60.6 / 62.3 / 65.2 ms per headline frame)
#endif

#ifndef PTX_POOL_WALK_KEEP
#define PTX_POOL_WALK_KEEP 5     // leave a WALK phase early once no more than KEEP/8 of the lanes it started with still walk (with any-hit
                                 // termination: 0 / 4 / 5 / 6 -> 75.5 / 73.9 / 73.2 / 73.2 ms per headline frame)
#endif

#ifndef PTX_POOL_TRI_PRIO
#define PTX_POOL_TRI_PRIO 3      // s_setprio of a wave while it runs a TRI phase (0 elsewhere)
#endif

#ifndef PTX_POOL_DEFER_CACHED
#define PTX_POOL_DEFER_CACHED 1  // 1: one pass through shade_step per SHADE phase (cached primary hits wait for the next phase)
#endif
#ifndef PTX_POOL_SKIP_PROBED
#define PTX_POOL_SKIP_PROBED 1   // the walk does not queue the leaf the blocker cache has queued already
#endif
#ifndef PTX_POOL_CACHE_RESET
#define PTX_POOL_CACHE_RESET 0   // 0: the blocker cache survives bounces and samples (63.6 -> 62.7 ms per headline frame: the next bounce is
                                 // often shadowed by the same object); a separate entry for bounce 0 (same point for every sample): no gain
#endif
#ifndef PTX_POOL_CACHE
#define PTX_POOL_CACHE 1          // 0: no blocker cache (diagnostic builds)
#endif
#ifndef PTX_POOL_CACHE2
#define PTX_POOL_CACHE2 1         // the blocker cache keeps the leaves of the last TWO different blockers (scenes with 256 .. 65 534 nodes: two 16-bit node
                                  // indices in one register).  Counted on the headline: a blocked light ray's blocker sits in the same leaf as the lane's
                                  // previous blocker in 59 % of the cases, in the last different leaf before that in 24 %, in the third / fourth 7 / 2 %.
                                  // Both leaves are probed when the ray starts and each one whose box is hit is queued: 10.9 -> 7.5 node visits and
                                  // 16.2 -> 15.1 triangle tests per traced ray, 60.7 -> 59.1 ms per headline frame (with SHADE_W 4: 58.4)
#endif
#ifndef PTX_POOL_MIN_WAVES
#define PTX_POOL_MIN_WAVES 5      // waves per SIMD the register allocation aims for (caps VGPRs at 96; LDS allows 20 waves per CU)
#endif

namespace ptx {

#define PTX_POOL_KEY_NONE ((unsigned long long)0x7e967699u << 32 | 0xFFFFFFFFull)   // bits(1.0e38f) << 32 | 0xFFFFFFFF: "no hit yet"

template <int PRUNE, bool MEMO, bool COUNT>
__global__ __launch_bounds__(64, PTX_POOL_MIN_WAVES) void k_render_pool(DScene S, DParams P, const int32_t* __restrict__ timestamps,
                                                    const float* __restrict__ prim_t, const int* __restrict__ prim_tri,
                                                    float4* __restrict__ fb, uint32_t n_frames, uint32_t chunk,
                                                    uint32_t* __restrict__ task_counter, unsigned long long* counters) {
    __shared__ float lds[PW_C_SAMPLES * 64];         // path state only (25 words per lane); work counters live in scalar registers
    __shared__ unsigned long long keys[64];          // best (t, slot) of the lane's ray
    __shared__ uint2 queue[PTX_POOL_QCAP];
    __shared__ uint32_t kap[64];                     // PRUNE = 1: packed kappa pair of the lane's ray (ray_kappa), read by whoever tests its leaves
    __shared__ float tlm[64];                        // PRUNE != 0: light_bound of the lane's ray if it is a light-visibility ray, else -3e38
    const uint32_t lane = threadIdx.x;
    const LdsLane L{lds + lane};
    const uint32_t npix = P.W * P.H;
    const uint32_t tiles_x = (P.W + 7) >> 3, tiles_y = (P.H + 7) >> 3;
    const uint32_t tasks_per_frame = tiles_x * tiles_y * 64u;          // padded to whole 8x8 tiles
    const uint32_t total = tasks_per_frame * n_frames;                 // host guarantees < 2^32
    // Persistent waves: the grid is one residency of the chip, and every wave draws `chunk` tasks at a time from a global
    // counter until the launch's task list is exhausted.  No wave runs dry while work is left, so lanes only idle during
    // the last chunk of the whole launch.
    uint32_t next = 0, last = 0;                                       // wave-uniform cursor into the current chunk
    bool exhausted = false;
    // set bits of a ballot below this lane: v_mbcnt_lo/hi (two instructions, no mask registers)
    auto below = [](unsigned long long m) { return (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)); };

    keys[lane] = PTX_POOL_KEY_NONE;
    unsigned long long cyc[3] = {0, 0, 0}, its[3] = {0, 0, 0};
    uint32_t cnt_nodes = 0, cnt_tris = 0, d_walk = 0, d_tri = 0, d_shade = 0, d_walk_s = 0, d_tri_s = 0, d_shade_s = 0;   // COUNT only

    int flags = F_ALIVE;
    V3 ro = mk(0.f, 0.f, 0.f), rd = mk(0.f, 0.f, 0.f), inv = mk(0.f, 0.f, 0.f);
    float limit = __builtin_inff();
    if (PRUNE) tlm[lane] = -3.0e38f;
    uint32_t i = 0xFFFFFFFFu, nbase = 0;
    const uint32_t N = (uint32_t)S.N;
    const float4* nodes_hi = S.nodes + 1;            // second half of a node record through its own base: no offset add
    uint32_t head = 0, tail = 0;                     // wave-uniform queue cursors (free running)
    // work counters as wave totals in scalar registers: every increment is a ballot + popcount in wave-uniform control flow
    uint32_t w_rays = 0, w_traced = 0, w_hits = 0, w_att = 0, w_pf = 0;
    uint32_t probed = 0xFFFFFFFFu;                   // fc word of the leaf the cache has already queued for the running ray: the walk skips it
    const bool cache2 = PTX_POOL_CACHE2 && (uint32_t)S.N >= 256u && (uint32_t)S.N < 0xFFFFu;   // (a walk through a tiny tree costs less than the second probe: my_cornell_6 +3 %)
    uint32_t c_noff = 0xFFFFFFFFu;                   // blocker cache: reference-tree node index of the last blocker's leaf; with cache2 two of them, 16 bits
                                                     // each (low half = newest, 0xFFFF = none)
    uint32_t lastpos = 0;                            // queue position after the lane's newest item: all of the ray's items have
                                                     // been tested once head has passed it (items are consumed in order)

    for (;;) {
        const bool trav = (flags & F_TRAV) != 0;
        if (PRUNE && trav && (uint32_t)(keys[lane] >> 32) == 0u) i = 0xFFFFFFFFu;   // settled by a conclusive blocker (distance bits 0): nothing left to walk
        const bool walking = trav && i < N;
        const bool need_shade = (flags & F_ALIVE) && !(trav && (i < N || (int32_t)(head - lastpos) < 0));
        const int nW = __popcll(__ballot(walking)), nS = __popcll(__ballot(need_shade));
        const uint32_t len = tail - head;
        if ((nW | nS) == 0 && len == 0) break;
        int phase;   // 0 walk, 1 tri, 2 shade
        if (len >= 64u) phase = 1;
        else {
            const int sw = nW, st = (int)len, ss = (nS * PTX_POOL_SHADE_W) >> 2;
            if (nS > 0 && ss >= sw && ss >= st) phase = 2;
            else if (nW > 0 && sw >= st) phase = 0;
            else phase = 1;
        }
        phase = __builtin_amdgcn_readfirstlane(phase);
        unsigned long long t_ph = 0;
        if (COUNT) t_ph = __builtin_amdgcn_s_memtime();

        // TRI phases are load-latency bound (five 16-B loads, then one burst of packed math): their instructions go first
        if (phase == 1) __builtin_amdgcn_s_setprio(PTX_POOL_TRI_PRIO); else __builtin_amdgcn_s_setprio(0);
        if (phase == 0) {
            Kappa kappa = {0.f, 0.f};
            if (PRUNE == 1) kappa = unpack_kappa(kap[lane]);
            bool w0 = walking;
            if (PRUNE) {
                const unsigned long long key = keys[lane];
                const float tb = __uint_as_float((uint32_t)(key >> 32));
                limit = prune_limit<PRUNE>(tb, PRUNE == 2 ? ray_lat(S, inv) : 0.0f);
                // any-hit rule for light-visibility rays: a TRI phase found a conclusive blocker (distance bits 0, below)
                if (tb == 0.0f) { i = 0xFFFFFFFFu; w0 = false; }
            }
            // blocker cache: before its first step a light ray tests the boxes of the leaves that blocked the lane's previous attempts and,
            // if hit, queues that leaf (the reference tests it too: its box passes, hence every ancestor's)
            if (PRUNE && __ballot(w0 && (flags & F_PROBE)) != 0ull) {
                bool probe = false, probe2 = false;
                uint32_t probe_item = 0, probe_item2 = 0;
                float probe_tn = 0.f, probe_tn2 = 0.f;
                if (w0 && (flags & F_PROBE)) {
                    // two entries (16-bit node indices, newest in the low half): both leaves' records are requested together, and
                    // each leaf whose box the ray hits is queued
                    const uint32_t n1 = cache2 ? (c_noff & 0xFFFFu) : c_noff, n2 = cache2 ? (c_noff >> 16) : 0xFFFFu;
                    const uint32_t o1 = S.ref_base + (n1 << 5), o2 = (cache2 && n2 != 0xFFFFu) ? S.ref_base + (n2 << 5) : o1;
                    const float4 a = ld16(S.nodes, o1), b = ld16(nodes_hi, o1);
                    float4 a2 = a, b2 = b;
                    if (cache2) { a2 = ld16(S.nodes, o2); b2 = ld16(nodes_hi, o2); }
                    probe = slab_node(ro, inv, a, b, probe_tn) && __float_as_int(b.w) >= 0;
                    probe_item = (uint32_t)__float_as_int(b.w) & 0x00FFFFFFu;
                    if (probe) probed = (uint32_t)__float_as_int(b.w);     // the leaf's fc word (first | count << 24): the walk skips it
                    if (cache2 && o2 != o1) {                               // (the older leaf may be queued again by the walk: tested twice, same minimum)
                        probe2 = slab_node(ro, inv, a2, b2, probe_tn2) && __float_as_int(b2.w) >= 0;
                        probe_item2 = (uint32_t)__float_as_int(b2.w) & 0x00FFFFFFu;
                    }
                    flags &= ~F_PROBE;
                }
                const unsigned long long m = __ballot(probe);     // at most 64 and len < 64 here
                if (probe) {
                    const uint32_t pos = tail + below(m);
                    queue[pos & (PTX_POOL_QCAP - 1)] = make_uint2(probe_item | (lane << 24), __float_as_uint(probe_tn));
                    lastpos = pos + 1u;
                }
                tail += (uint32_t)__popcll(m);
                const unsigned long long m2 = __ballot(probe2);
                if (m2 != 0ull && tail - head + (uint32_t)__popcll(m2) <= (uint32_t)PTX_POOL_QCAP) {   // room for all of them, or none
                    if (probe2) {
                        const uint32_t pos = tail + below(m2);
                        queue[pos & (PTX_POOL_QCAP - 1)] = make_uint2(probe_item2 | (lane << 24), __float_as_uint(probe_tn2));
                        lastpos = pos + 1u;
                    }
                    tail += (uint32_t)__popcll(m2);
                }
            }
            // ordered = every walking lane may take the entry-first shortcut (octant layouts, finite 1/d): wave-uniform
            const bool ordered = S.oct_stride != 0u && __ballot(w0 && (flags & F_INF)) == 0ull;
            auto walk = [&](auto MODE) {
                bool w = w0;                                      // trav && i < N, carried from one step to the next
#pragma nounroll
                for (int g = 0; g < PTX_POOL_WALK_STEPS; g++) {
                    if (tail - head > (uint32_t)(PTX_POOL_QCAP - 64)) break;
                    bool push = false;
                    uint32_t item = 0;
                    float tn = 0.f;
                    if (w) {
                        const float4 a = ld16(S.nodes, nbase + (i << 5));
                        const float4 b = ld16(nodes_hi, nbase + (i << 5));
                        int miss = __float_as_int(b.z), fc = __float_as_int(b.w);
                        asm volatile("" : "+v"(miss), "+v"(fc));     // keep the links in the record's two loads (no third, dependent one)
                        if (COUNT) cnt_nodes++;
                        // MODE 0: every walking lane takes the entry-first shortcut; 1: lanes with an infinite 1/d component walk
                        // the reference tree with the selecting test (second-operand NaN rule), the others stay entry-first;
                        // 2: single-layout views (the reference tree for everyone): selecting tests only
                        // (on finite slabs the selecting test with either NaN rule IS the entry-first test, so in mode 1 one
                        // instruction stream serves both kinds of lane instead of two divergent ones)
                        const bool hit0 = decltype(MODE)::value == 0 ? slab_node_ordered(ro, inv, a, b, tn)
                                        : (decltype(MODE)::value == 1 ? slab_node_ref(ro, inv, a, b, tn)
                                        : ((flags & F_INF) ? slab_node_ref(ro, inv, a, b, tn) : slab_node(ro, inv, a, b, tn)));
                        bool hit = hit0;
                        // a NaN or partial tn is still a lower bound of the entry distance
                        if (PRUNE == 2) hit = hit && !(tn > limit);
                        if (PRUNE == 1) hit = hit && !(fc < 0 && prune_far<1>(tn, limit, (uint32_t)fc, kappa));   // leaves: at dequeue, with their own coefficients
                        push = hit && fc >= 0 && (!PTX_POOL_SKIP_PROBED || (uint32_t)fc != probed);   // internal nodes and empty leaves carry -2: nothing to queue
                        item = (uint32_t)fc;
                        i = hit ? i + 1u : (uint32_t)miss;
                    }
                    const unsigned long long m = __ballot(push);
                    if (push) {
                        const uint32_t pos = tail + below(m);
                        queue[pos & (PTX_POOL_QCAP - 1)] = make_uint2((item & 0x00FFFFFFu) | (lane << 24), __float_as_uint(tn));
                        lastpos = pos + 1u;
                    }
                    tail += (uint32_t)__popcll(m);
                    if (COUNT) { d_walk += w ? 1u : 0u; if (lane == 0) d_walk_s += 64u; }
                    w = trav && i < N;
                    if (__popcll(__ballot(w)) * 8 <= nW * PTX_POOL_WALK_KEEP) break;   // KEEP = 0: only when nobody walks
                }
            };
            if (ordered) walk(std::integral_constant<int, 0>{});
            else if (S.oct_stride != 0u) walk(std::integral_constant<int, 1>{});
            else walk(std::integral_constant<int, 2>{});
        } else if (phase == 1) {
            const uint32_t n = min(len, 64u);
            const bool has = lane < n;
            const uint2 it = queue[(head + lane) & (PTX_POOL_QCAP - 1)];
            head += n;
            const uint32_t owner = has ? ((it.x >> 24) & 63u) : lane;
            const uint32_t first = it.x & 0x00FFFFFFu;
            // the first pair record of the leaf is requested at once, so its latency overlaps the LDS traffic below
            uint32_t pbyte = first * 80u;
            float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0, r4 = r0;
            if (has) {
                r0 = ld16(S.pairs, pbyte); r1 = ld16(S.pairs, pbyte + 16u); r2 = ld16(S.pairs, pbyte + 32u);
                r3 = ld16(S.pairs, pbyte + 48u); r4 = ld16(S.pairs, pbyte + 64u);
            }
            // the owner's ray, straight out of the owner's registers (every lane executes the shuffles)
            const V3 o = mk(__shfl(ro.x, (int)owner), __shfl(ro.y, (int)owner), __shfl(ro.z, (int)owner));
            const V3 d = mk(__shfl(rd.x, (int)owner), __shfl(rd.y, (int)owner), __shfl(rd.z, (int)owner));
            bool go = has;
            if (PRUNE) {
                const float b = __uint_as_float((uint32_t)(keys[owner] >> 32));
                Kappa ok = {0.f, 0.f};
                if (PRUNE == 1) ok = unpack_kappa(kap[owner]);
                go = go && !prune_far<PRUNE>(__uint_as_float(it.y), prune_limit<PRUNE>(b, PRUNE == 2 ? ray_lat_d(S, d) : 0.0f), __float_as_uint(r4.w), ok);   // r4.w: the leaf's packed coefficients
            }
            float bt = 1.0e38f;
            uint32_t bk = 0xFFFFFFFFu;
            uint32_t k = first;                               // slot of triangle A of the pair; the pair record is pairs[first + round]
            bool more = go;
            while (__ballot(more) != 0ull) {
                if (COUNT) { d_tri += more ? 1u : 0u; if (lane == 0) d_tri_s += 64u; }
                if (more) {
                    const int rz = __float_as_int(r4.z);
                    const int rem = rz & 0xFF;                 // triangles of the leaf from this pair on (>= 1); bits 16 / 17: A / B emit
                    bool hA, hB;
                    float tA, tB;
                    tri_test2(o, d, r0, r1, r2, r3, r4, hA, hB, tA, tB);
                    if (COUNT) cnt_tris += rem >= 2 ? 2u : 1u;
                    if (hA && tA < bt) { bt = tA; bk = (k << 1) | ((uint32_t)(rz >> 16) & 1u); }
                    if (hB && tB < bt) { bt = tB; bk = ((k + 1u) << 1) | ((uint32_t)(rz >> 17) & 1u); }
                    more = rem > 2;
                    k += 2u;
                    if (more) {
                        pbyte += 80u;
                        r0 = ld16(S.pairs, pbyte); r1 = ld16(S.pairs, pbyte + 16u); r2 = ld16(S.pairs, pbyte + 32u);
                        r3 = ld16(S.pairs, pbyte + 48u); r4 = ld16(S.pairs, pbyte + 64u);
                    }
                }
            }
            if (bt < 1.0e38f) {
                // any-hit rule for light-visibility rays: a hit on a triangle without emission, nearer than anything emissive can
                // be (light_bound of the owner's ray; -3e38 for other rays), settles the ray: stored with distance 0 it is "a
                // hit", ends the owner's walk at its next WALK phase and puts every leaf still queued out of reach
                if (PRUNE && S.n_lbox != 0u && (bk & 1u) == 0u) {
                    Kappa ok = {0.f, 0.f};
                    if (PRUNE == 1) ok = unpack_kappa(kap[owner]);
                    if (light_settles<PRUNE>(S, tlm[owner], bt, ok)) bt = 0.0f;
                }
                atomicMin(&keys[owner], ((unsigned long long)__float_as_uint(bt) << 32) | (unsigned long long)bk);
            }
        } else {
            if (COUNT) { d_shade += need_shade ? 1u : 0u; if (lane == 0) d_shade_s += 64u; }
            bool run_shade = need_shade && (flags & F_TASK);      // a finished ray (or a pending cached primary hit)
            bool want_task = need_shade && !(flags & F_TASK);
            bool start_trav = false;
            PathLds p{L, ro, rd, P, prim_t, prim_tri};
            // (1) hand out the wave's next tasks (wave-uniform control flow: the cursor `next` stays uniform)
            for (;;) {
                const unsigned long long m = __ballot(want_task);
                if (m == 0) break;
                if (next == last && !exhausted) {                          // wave-uniform: draw the next chunk
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(task_counter, chunk);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    if (base >= total) exhausted = true;
                    else { next = base; last = (total - base > chunk) ? base + chunk : total; }
                }
                const uint32_t mine = next + below(m);
                const uint32_t avail = last - next;
                next += min((uint32_t)__popcll(m), avail);
                bool black = false;
                if (want_task) {
                    if (mine >= last) {
                        if (exhausted) { flags = 0; want_task = false; }   // else: draws from the next chunk in the next round
                    } else {
                        // task list: tile-major, frame-minor, the 64 pixels of a tile together: concurrently running waves work on
                        // the same and on neighbouring tiles (measured against frame-major order: 93.9 -> 85.3 ms/frame at 4 frames
                        // per launch, 83.4 -> 81.3 at 32)
                        const uint32_t c64 = mine >> 6;
                        const uint32_t tile = c64 / n_frames;
                        const uint32_t frame = c64 - tile * n_frames;
                        uint32_t px, py;
                        if (tile_pixel(P, tile, mine & 63u, px, py)) {     // else: padding lane of an edge tile, draw again
                            const uint32_t pix = py * P.W + px;
                            L.sti(PW_PIX, (int)(pix | (frame << 26))); L.sti(PW_PACK, 0);
                            LocalCnt cb = {};
                            const int st = path_begin<MEMO>(P, p, px, py, timestamps[frame], 0.f, 0, cb);
                            if (st == SH_DONE) {                            // max_path_length <= 0: black; draw again
                                fb[(size_t)frame * npix + pix] = make_float4(0.f, 0.f, 0.f, 1.0f);
                                black = true;
                            } else {
                                flags |= F_TASK;
                                want_task = false;
                                if (st == SH_CACHED) { flags |= F_CACHED; run_shade = true; }
                                else { flags &= ~F_CACHED; start_trav = true; }
                            }
                        }
                    }
                }
                w_pf += (uint32_t)__popcll(__ballot(black));
            }
            // (2) consume finished rays / cached primary hits.  A lane may go through several samples in a row (primary ray
            //     misses or hits an emitter: every sample ends at once); the loop is wave-uniform so the counters stay scalar
            {
                float t = 0.f;
                bool hit = false;
                int tri = -1;
                bool cached = (flags & F_CACHED) != 0;
                if (run_shade && !cached) {
                    const unsigned long long key = keys[lane];
                    t = __uint_as_float((uint32_t)(key >> 32));
                    hit = t < 1.0e38f;
                    if (hit) {
                        const uint32_t slot = (uint32_t)key >> 1;
                        if (p.stage() != ST_PATH && ((uint32_t)key & 1u) == 0u) {
                            // a light-visibility ray stopped by a triangle without emission: shade_step needs no index (tri < 0);
                            // the blocker's leaf is remembered for the next attempt instead
                            tri = -1;
                            const uint32_t nn = (uint32_t)__float_as_int(ld16(S.tris, slot * 48u + 32u).w);   // reference-tree node of the blocker's leaf
                            if (!cache2) c_noff = nn;
                            else if (nn != (c_noff & 0xFFFFu)) c_noff = (c_noff << 16) | nn;
                        } else {
                            tri = __float_as_int(ld16(S.tris, slot * 48u).w);
                        }
                    }
                }
                bool act = run_shade;
                int st = SH_TRACE;
                // PTX_POOL_DEFER_CACHED: a lane whose sample ended meets its cached primary hit in the NEXT shade phase, together
                // with the other lanes that have a hit to process, instead of a second pass through shade_step for a few lanes now
                for (int pass = 0; (PTX_POOL_DEFER_CACHED ? pass == 0 : true) && __ballot(act) != 0ull; pass++) {
                    LocalCnt cc = {};
                    if (act) {
                        if (cached) { t = p.prim_t(); hit = t < 1.0e38f; tri = p.prim_tri(); }
                        st = shade_step<MEMO>(S, P, p, hit, t, tri, cc);
                        cached = (st == SH_CACHED);
                        act = cached;
                    }
                    // per call: rays 1 or 2 (the memoised shadow ray), hits 0..2, light attempts 0 or 1
                    w_rays += (uint32_t)__popcll(__ballot(cc.rays >= 1u)) + (uint32_t)__popcll(__ballot(cc.rays >= 2u));
                    w_hits += (uint32_t)__popcll(__ballot(cc.ray_hits >= 1u)) + (uint32_t)__popcll(__ballot(cc.ray_hits >= 2u));
                    w_att += (uint32_t)__popcll(__ballot(cc.light_attempts >= 1u));
                }
                bool done = false;
                if (run_shade && st == SH_CACHED) {            // (deferred) needs shade again: no ray in flight
                    flags = (flags | F_CACHED) & ~F_TRAV;
                } else if (run_shade) {
                    flags &= ~F_CACHED;
                    if (st == SH_DONE) {
                        const V3 rgb = path_result(P, p);
                        fb[(size_t)p.frame() * npix + p.pix()] = make_float4(rgb.x, rgb.y, rgb.z, 1.0f);
                        done = true;
                        flags &= ~(F_TASK | F_TRAV);          // takes its next task in the next SHADE phase
                    } else {
                        start_trav = true;
                    }
                }
                w_pf += (uint32_t)__popcll(__ballot(done));
            }
            // (3) start the traversal of the lanes that now hold a fresh ray
            if (start_trav) {
                inv = mk(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
                keys[lane] = PTX_POOL_KEY_NONE;
                uint32_t pk = 0u;
                if (PRUNE == 1) { pk = pack_kappa(ray_kappa(S, ro, rd)); kap[lane] = pk; }
                limit = __builtin_inff();
                i = 0;
                lastpos = head;
                // rays with an infinite or NaN 1/d component walk the reference tree with the selecting slab test (slab_node_ref)
                const bool deg = inv_degenerate(inv);
                flags = (flags & ~(F_INF | F_PROBE)) | F_TRAV | (deg ? F_INF : 0);
                probed = 0xFFFFFFFFu;
                nbase = deg ? S.ref_base : octant(rd) * S.oct_stride;
                const bool lightray = p.stage() != ST_PATH;
#if PTX_POOL_CACHE_RESET
                if (!lightray) c_noff = 0xFFFFFFFFu;         // 1: the cache belongs to the attempts of one bounce
#endif
                if (PRUNE && S.n_lbox != 0u) {               // glsl:353-363 / 439-442: only "hit" and the closest hit's emission are used
                    tlm[lane] = (lightray && !deg) ? light_bound<PRUNE>(S, ro, rd, inv, unpack_kappa(pk)) : -3.0e38f;
                    if (PTX_POOL_CACHE && lightray && !deg && c_noff != 0xFFFFFFFFu) flags |= F_PROBE;   // blocker cache: tested when the walk starts
                }
            }
            w_traced += (uint32_t)__popcll(__ballot(start_trav));
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
    if (lane == 0) {   // every task runs exactly spp samples (glsl:521-525), so samples = pixel-frames x spp
        c.samples = w_pf * (uint32_t)P.spp; c.rays = w_rays; c.rays_traced = w_traced; c.ray_hits = w_hits; c.light_attempts = w_att;
    }
    c.node_visits = cnt_nodes; c.tri_tests = cnt_tris;
    c.walk_useful = d_walk; c.walk_slots = d_walk_s; c.tri_useful = d_tri; c.tri_slots = d_tri_s; c.trace_useful = d_shade; c.trace_slots = d_shade_s;
    flush_counters(c, counters, lane == 0 ? w_pf : 0u);
}

} // namespace ptx
