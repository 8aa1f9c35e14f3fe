// ptx_kernels.h -- device data layout + kernels of the progressive path tracer (gfx950).
//
// HBM layout (replaces the eleven RGB32F data textures of main.js:272-282; DESIGN.md "Data layout"):
//   nodeA[N]  float4 { min.xyz, bits(miss link) }
//   nodeB[N]  float4 { max.xyz, bits(leaf ? first | count << 24 : -2) }
//   triV0[T]  float4 { v0.xyz, bits(original triangle index) }   -- indexed by INORDER SLOT k, so a
//   triE1[T]  float4 { (v1 - v0).xyz, 0 }                             leaf's triangles are contiguous
//   triE2[T]  float4 { (v2 - v0).xyz, 0 }
//   attrN/attrC/attrE[T] float4 { normal | color | emission } by ORIGINAL triangle index
//   light[L]  6 x float4 { v0, v1, v2, normal, emission, {1/(L*area),0,0,0} }
// One node visit = 32 B, one triangle test = 48 B, attributes (48 B) only for the winning hit.
#pragma once
#include "ptx_device.h"

namespace ptx {

struct DScene {
    const float4* nodeA;
    const float4* nodeB;
    const float4* triV0;
    const float4* triE1;
    const float4* triE2;
    const float4* attrN;
    const float4* attrC;
    const float4* attrE;
    const float4* light;   // 6 float4 per light
    int N, T, L;
};

struct DParams {
    uint32_t W, H;
    V3 src, up, right, lb;
    float aspect;
    int max_len, spp;
};

// device-side work counters (u64), one block of these per context
enum { CNT_SAMPLES = 0, CNT_RAYS, CNT_RAYS_TRACED, CNT_NODE_VISITS, CNT_TRI_TESTS, CNT_RAY_HITS, CNT_LIGHT_ATTEMPTS, CNT_PIXEL_FRAMES, CNT_NUM };

struct LocalCnt {
    uint32_t samples, rays, rays_traced, node_visits, tri_tests, ray_hits, light_attempts;
};

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
}

// Margin of the best-t pruning (option "traversal" = 1): a node is skipped only when its slab entry
// distance exceeds best_t * (1 + PRUNE_REL) + PRUNE_ABS.  See DESIGN.md "Pruning and exactness".
#define PTX_PRUNE_REL 0.001f
#define PTX_PRUNE_ABS 0.001f

// glsl:207-249: closest hit by the stackless threaded traversal (hit -> i+1, miss -> missLink).
// Strict '<' keeps the first candidate in inorder position among equal t (the reference's tie rule).
// PRUNE adds the conservative best-t cut; COUNT adds node/triangle counters.
template <bool PRUNE, bool COUNT>
__device__ __forceinline__ bool closest_hit(const DScene& S, V3 o, V3 d, float& out_t, int& out_tri, LocalCnt& c) {
    const V3 inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    float best = 1.0e38f;
    int best_tri = -1;
    int i = 0;
    const int N = S.N;
    float limit = 3.0e38f;
    while (i != -1 && i < N) {
        const float4 a = S.nodeA[i];
        const float4 b = S.nodeB[i];
        if (COUNT) c.node_visits++;
        float tn;
        bool hit = slab_test(o, inv, mk(a), mk(b), tn);
        if (PRUNE) hit = hit && !(tn > limit);
        if (hit) {
            const int fc = __float_as_int(b.w);
            if (fc != -2) {
                const int first = fc & 0x00FFFFFF, cnt = (fc >> 24) & 0x7F;
                for (int k = first; k < first + cnt; k++) {
                    const float4 v0 = S.triV0[k];
                    const float4 e1 = S.triE1[k];
                    const float4 e2 = S.triE2[k];
                    if (COUNT) c.tri_tests++;
                    float t;
                    if (tri_test(o, d, mk(v0), mk(e1), mk(e2), t) && t < best) {
                        best = t;
                        best_tri = __float_as_int(v0.w);
                        if (PRUNE) limit = best * (1.0f + PTX_PRUNE_REL) + PTX_PRUNE_ABS;
                    }
                }
            }
            i++;
        } else {
            i = __float_as_int(a.w);
        }
    }
    out_t = best;
    out_tri = best_tri;
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
