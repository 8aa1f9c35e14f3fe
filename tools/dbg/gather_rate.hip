// gather_rate.hip -- how many scattered vector loads per cycle does a gfx950 CU sustain from L1 / L2?
//
// Second ceiling of DESIGN.md "Roofline": the path tracer issues 1.8e9 wave-level `global_load_dwordx4` per headline frame,
// nearly all of them gathers (every lane its own 16-B piece of a node or triangle-pair record of a ~5 MB scene), and the
// texture-address unit is 89 % busy.  This program measures what such loads cost by themselves: single-wave workgroups,
// W waves on every SIMD (the renderer runs 5), each wave issuing a long stream of independent loads whose per-lane addresses
// come from an LCG (so they do not depend on loaded data), `inflight` loads outstanding per wave.
//   shape 0: every lane a random 16-B slot of the region           (dwordx4)
//   shape 1: every lane a random 32-B record, both halves          (2 x dwordx4: a node visit)
//   shape 2: every lane a random 80-B record, all five pieces      (5 x dwordx4: a triangle-pair test)
//   shape 3: 64 consecutive 16-B slots at a random 1-KiB-aligned place (coalesced, for reference)
//   shape 4: every lane a random 4-B word                          (dword)
//   shape 5: every lane a random 8-B word                          (dwordx2)
//   shape 6: like 1, but only `active` of 64 lanes enabled          (what a thinning WALK phase issues)
//   shape 7: groups of g adjacent lanes read g x 16 contiguous bytes at a random (g x 16)-aligned place (g = `active`)
//   shape 8: groups of g adjacent lanes read the SAME random 16-B slot (coherent rays)
//   shape 10: runs of 4 adjacent lanes read 4 adjacent 48-B records (a leaf's triangles), 3 x dwordx4 each
//   shape 11: runs of 4 adjacent lanes read 4 adjacent 36-B records, 3 x dwordx3 each
//   shape 12: runs of 4 adjacent lanes read a 192-B block piece-major (3 x dwordx4, each quad 64 contiguous bytes per load)
//   shape 9: like 0, but only every (64 / active)-th lane enabled   (active lanes spread over the wave)
// Output: one JSON line per case: wave-instructions per cycle per CU (at the shader clock measured with s_memtime against
// wall_clock64) and bytes per cycle per CU.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/dbg/gather_rate tools/dbg/gather_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct F3 { float x, y, z; };
struct Stamp { unsigned long long t0, t1, w0, w1; };

__device__ __forceinline__ float sum4(const float4 v) { return (v.x + v.y) + (v.z + v.w); }
__device__ __forceinline__ uint32_t lcg(uint32_t& x) { x = x * 1664525u + 1013904223u; return x; }

template <int SHAPE>
__global__ __launch_bounds__(64) void k_gather(const float4* __restrict__ base, uint32_t region_bytes, int iters, int active, Stamp* st, float* sink) {
    uint32_t x = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
    float acc = 0.0f;
    const unsigned long long w0 = wall_clock64();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (SHAPE >= 10) x = ((blockIdx.x * 64u + threadIdx.x / 4u) * 2654435761u + 12345u);
    if (SHAPE == 7 || SHAPE == 8) x = ((blockIdx.x * 64u + threadIdx.x / (uint32_t)active) * 2654435761u + 12345u);
    const bool on = SHAPE == 6 ? (int)threadIdx.x < active : SHAPE == 9 ? (threadIdx.x % (64u / (uint32_t)active)) == 0u : true;
    if (on) {
        for (int i = 0; i < iters; i++) {
            if (SHAPE == 0) {
                const uint32_t n = region_bytes / 16u;
#pragma unroll
                for (int k = 0; k < 8; k++) { const float4 v = base[__umulhi(lcg(x), n)]; acc += sum4(v); }
            } else if (SHAPE == 1 || SHAPE == 6) {
                const uint32_t n = region_bytes / 32u;
#pragma unroll
                for (int k = 0; k < 4; k++) { const uint32_t r = (__umulhi(lcg(x), n)) * 2u; const float4 a = base[r], b = base[r + 1]; acc += sum4(a) + sum4(b); }
            } else if (SHAPE == 2) {
                const uint32_t n = region_bytes / 80u;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t r = (__umulhi(lcg(x), n)) * 5u;
                    const float4 a = base[r], b = base[r + 1], c = base[r + 2], d = base[r + 3], e = base[r + 4];
                    acc += (sum4(a) + sum4(b)) + (sum4(c) + sum4(d)) + sum4(e);
                }
            } else if (SHAPE == 3) {
                const uint32_t n = region_bytes / 1024u;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const uint32_t r = __umulhi(__builtin_amdgcn_readfirstlane(lcg(x)), n);
                    const float4 v = base[r * 64u + threadIdx.x]; acc += sum4(v);
                }
            } else if (SHAPE == 7) {
                const uint32_t n = region_bytes / (16u * (uint32_t)active);
#pragma unroll
                for (int k = 0; k < 8; k++) { const float4 v = base[__umulhi(lcg(x), n) * (uint32_t)active + threadIdx.x % (uint32_t)active]; acc += sum4(v); }
            } else if (SHAPE == 8 || SHAPE == 9) {
                const uint32_t n = region_bytes / 16u;
#pragma unroll
                for (int k = 0; k < 8; k++) { const float4 v = base[__umulhi(lcg(x), n)]; acc += sum4(v); }
            } else if (SHAPE == 10) {
                const uint32_t n = region_bytes / 192u;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t r = (__umulhi(lcg(x), n) * 4u + (threadIdx.x & 3u)) * 3u;
                    const float4 a = base[r], b = base[r + 1], c = base[r + 2];
                    acc += (sum4(a) + sum4(b)) + sum4(c);
                }
            } else if (SHAPE == 11) {
                const uint32_t n = region_bytes / 144u;
                const F3* fb = reinterpret_cast<const F3*>(base);
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t r = (__umulhi(lcg(x), n) * 4u + (threadIdx.x & 3u)) * 3u;
                    const F3 a = fb[r], b = fb[r + 1], c = fb[r + 2];
                    acc += ((a.x + a.y) + (a.z + b.x)) + ((b.y + b.z) + (c.x + c.y)) + c.z;
                }
            } else if (SHAPE == 12) {
                const uint32_t n = region_bytes / 192u;
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const uint32_t r = __umulhi(lcg(x), n) * 12u + (threadIdx.x & 3u);
                    const float4 a = base[r], b = base[r + 4], c = base[r + 8];
                    acc += (sum4(a) + sum4(b)) + sum4(c);
                }
            } else if (SHAPE == 4) {
                const uint32_t n = region_bytes / 4u;
#pragma unroll
                for (int k = 0; k < 8; k++) acc += reinterpret_cast<const float*>(base)[__umulhi(lcg(x), n)];
            } else if (SHAPE == 5) {
                const uint32_t n = region_bytes / 8u;
#pragma unroll
                for (int k = 0; k < 8; k++) { const float2 v = reinterpret_cast<const float2*>(base)[__umulhi(lcg(x), n)]; acc += v.x + v.y; }
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{t0, t1, w0, w1};
    sink[blockIdx.x * 64 + threadIdx.x] = acc;
}

static int loads_per_iter(int shape) { return shape == 2 ? 10 : (shape >= 10 ? 6 : 8); }
static int bytes_per_lane_load(int shape) { return shape == 4 ? 4 : shape == 5 ? 8 : shape == 11 ? 12 : 16; }

template <int SHAPE>
static int run(const float4* d_base, uint32_t region, int waves_per_simd, int iters, int active, Stamp* d_st, float* d_sink, int n_cu) {
    const int grid = n_cu * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<SHAPE>, dim3(grid), dim3(64), 0, 0, d_base, region, 16, active, d_st, d_sink);   // warm caches
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<SHAPE>, dim3(grid), dim3(64), 0, 0, d_base, region, iters, active, d_st, d_sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> st(grid);
    CHECK(hipMemcpy(st.data(), d_st, sizeof(Stamp) * grid, hipMemcpyDeviceToHost));
    // shader clock: memtime ticks per wall second (wall_clock64 = 100 MHz), averaged over waves; span = median wave duration
    double ticks = 0, wall = 0; unsigned long long tmin = ~0ull, tmax = 0;
    for (const Stamp& s : st) { ticks += double(s.t1 - s.t0); wall += double(s.w1 - s.w0); if (s.t0 < tmin) tmin = s.t0; if (s.t1 > tmax) tmax = s.t1; }
    const double memtime_hz = ticks / (wall / 100e6);
    const double wave_instr = double(grid) * iters * loads_per_iter(SHAPE);
    const double clock_ghz = 2.4;                       // s_memtime runs at a fixed 100 MHz on this part; use the nominal shader clock
    const double cycles = ms * 1e-3 * clock_ghz * 1e9;
    const double per_cu = wave_instr / n_cu / cycles;
    const int lanes = (SHAPE == 6 || SHAPE == 9) ? active : 64;
    printf("{\"shape\": %d, \"region_kib\": %u, \"waves_per_simd\": %d, \"active_lanes\": %d, \"ms\": %.3f, \"wave_loads_per_cycle_per_cu\": %.5f, "
           "\"cycles_per_wave_load\": %.2f, \"bytes_per_cycle_per_cu\": %.2f, \"chip_tb_per_s\": %.2f, \"memtime_mhz\": %.1f}\n",
           SHAPE, region / 1024u, waves_per_simd, lanes, ms, per_cu, 1.0 / per_cu, per_cu * lanes * bytes_per_lane_load(SHAPE),
           per_cu * lanes * bytes_per_lane_load(SHAPE) * n_cu * clock_ghz * 1e9 / 1e12, memtime_hz / 1e6);
    fflush(stdout);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int n_cu = prop.multiProcessorCount;
    const size_t cap = 64u << 20;
    float4* d_base; CHECK(hipMalloc(&d_base, cap)); CHECK(hipMemset(d_base, 0, cap));
    Stamp* d_st; CHECK(hipMalloc(&d_st, sizeof(Stamp) * n_cu * 4 * 8));
    float* d_sink; CHECK(hipMalloc(&d_sink, sizeof(float) * 64 * n_cu * 4 * 8));
    const uint32_t regions[] = {16u << 10, 1u << 20, 5u << 20, 32u << 20};
    for (uint32_t region : regions) {
        for (int w : {5, 8}) {
            if (run<0>(d_base, region, w, iters, 64, d_st, d_sink, n_cu)) return 1;
            if (run<1>(d_base, region, w, iters, 64, d_st, d_sink, n_cu)) return 1;
            if (run<2>(d_base, region, w, iters, 64, d_st, d_sink, n_cu)) return 1;
        }
        if (run<3>(d_base, region, 5, iters, 64, d_st, d_sink, n_cu)) return 1;
        if (run<4>(d_base, region, 5, iters, 64, d_st, d_sink, n_cu)) return 1;
        if (run<5>(d_base, region, 5, iters, 64, d_st, d_sink, n_cu)) return 1;
    }
    for (int active : {48, 32, 16, 8}) if (run<6>(d_base, 5u << 20, 5, iters, active, d_st, d_sink, n_cu)) return 1;
    for (uint32_t region : {16u << 10, 1u << 20, 5u << 20}) {
        if (run<10>(d_base, region, 5, iters, 64, d_st, d_sink, n_cu)) return 1;
        if (run<11>(d_base, region, 5, iters, 64, d_st, d_sink, n_cu)) return 1;
        if (run<12>(d_base, region, 5, iters, 64, d_st, d_sink, n_cu)) return 1;
    }
    if (argc > 2) return 0;
    for (uint32_t region : {16u << 10, 5u << 20}) {
        for (int g : {2, 4, 8, 16}) if (run<7>(d_base, region, 5, iters, g, d_st, d_sink, n_cu)) return 1;
        for (int g : {2, 4, 8, 16, 64}) if (run<8>(d_base, region, 5, iters, g, d_st, d_sink, n_cu)) return 1;
        for (int a : {32, 16, 8}) if (run<9>(d_base, region, 5, iters, a, d_st, d_sink, n_cu)) return 1;
        for (int a : {32, 16, 8}) if (run<6>(d_base, region, 5, iters, a, d_st, d_sink, n_cu)) return 1;
    }
    return 0;
}
