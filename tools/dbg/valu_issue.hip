// valu_issue.hip -- what does one wave64 VALU instruction cost a gfx950 SIMD, at 1 .. 8 waves per SIMD?
//
// Settles the issue cost behind bench.py's `roofline.valu` block (DESIGN.md "Roofline"; VERDICT r01 weak #1: the guide's
// constants table lists 2 cycles for wave64 v_fma_f32 on the SIMD-32 when several waves are resident and 4 for one wave
// alone).  Each kernel runs a long stream of INDEPENDENT instructions of one kind (8 accumulators, .rept-unrolled inline
// asm, so the compiler cannot change the mix) in single-wave workgroups; the grid is W waves for each of the chip's SIMDs.
// Every wave stamps s_memtime at both ends and reports where it ran (HW_REG_HW_ID / HW_REG_XCC_ID), so the host can
//   * check that the waves really were spread W per SIMD, and
//   * compute, per SIMD, cycles per wave-instruction = (last end - first start) / (instructions issued on that SIMD).
// wall_clock64() (100 MHz) is stamped too: ticks of s_memtime per wall second = the shader clock under this load.
//
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/dbg/valu_issue tools/dbg/valu_issue.hip      Run: tools/dbg/valu_issue > out.json
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Stamp { unsigned long long t0, t1, w0, w1; uint32_t hw_id, xcc_id; uint32_t pad[2]; };

#define REPS 16            // .rept count: 8 x 16 = 128 instructions per loop iteration
#define PER_ITER (8 * REPS)

// op3: "op d, a, b, c"-style with d = accumulator.  The 8 accumulators are independent chains; consecutive instructions
// never depend on each other (distance 8), so issue, not latency, is measured.
#define STREAM3(name, line)                                                                                        \
    __global__ __launch_bounds__(64) void name(Stamp* st, float* sink, int iters) {                                \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = 1.0000001f, c = 0.5f;                                                                            \
        asm volatile("" : "+v"(b), "+v"(c));                                                                       \
        const unsigned long long w0 = wall_clock64();                                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                \
        for (int i = 0; i < iters; i++)                                                                            \
            asm volatile(".rept 16\n" line(%0) line(%1) line(%2) line(%3) line(%4) line(%5) line(%6) line(%7) ".endr\n" \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        __builtin_amdgcn_s_waitcnt(0);                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                \
        const unsigned long long w1 = wall_clock64();                                                              \
        if (threadIdx.x == 0) {                                                                                    \
            Stamp s = {t0, t1, w0, w1, (uint32_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)),        \
                       (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)), {0, 0}};               \
            st[blockIdx.x] = s;                                                                                    \
        }                                                                                                          \
        sink[blockIdx.x * 64 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));                   \
    }

#define L_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define L_MUL(r) "v_mul_f32 " #r ", " #r ", %8\n"
#define L_ADD(r) "v_add_f32 " #r ", " #r ", %9\n"
#define L_MAX(r) "v_max_f32 " #r ", " #r ", %9\n"
#define L_RCP(r) "v_rcp_f32 " #r ", " #r "\n"
#define L_SQRT(r) "v_sqrt_f32 " #r ", " #r "\n"
#define L_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8\n"
#define L_CNDMASK(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc\n"
#define L_CMP(r) "v_cmp_lt_f32 vcc, " #r ", %8\n"
#define L_MOV(r) "v_mov_b32 " #r ", %8\n"

STREAM3(k_fma, L_FMA)
STREAM3(k_mul, L_MUL)
STREAM3(k_add, L_ADD)
STREAM3(k_max, L_MAX)
STREAM3(k_rcp, L_RCP)
STREAM3(k_sqrt, L_SQRT)
STREAM3(k_mullo, L_MULLO)
STREAM3(k_cndmask, L_CNDMASK)
STREAM3(k_cmp, L_CMP)
STREAM3(k_mov, L_MOV)
// second batch: the rest of the renderer's instruction mix
#define L_MIN(r) "v_min_f32 " #r ", " #r ", %9\n"
#define L_MAX3(r) "v_max3_f32 " #r ", " #r ", %8, %9\n"
#define L_MIN3(r) "v_min3_f32 " #r ", " #r ", %8, %9\n"
#define L_SUB(r) "v_sub_f32 " #r ", " #r ", %9\n"
#define L_AND(r) "v_and_b32 " #r ", " #r ", %8\n"
#define L_XOR(r) "v_xor_b32 " #r ", " #r ", %8\n"
#define L_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r "\n"
#define L_ADDU(r) "v_add_u32 " #r ", " #r ", %8\n"
#define L_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9\n"
#define L_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", %8\n"
#define L_CVT(r) "v_cvt_f32_i32 " #r ", " #r "\n"
#define L_BFE(r) "v_bfe_u32 " #r ", " #r ", 3, 8\n"
#define L_CND64(r) "v_cndmask_b32_e64 " #r ", " #r ", %8, s[20:21]\n"
#define L_CMPCND(r) "v_cmp_lt_f32 vcc, " #r ", %8\n v_cndmask_b32 " #r ", " #r ", %9, vcc\n"
#define L_CMPS(r) "v_cmp_lt_f32 s[20:21], " #r ", %8\n"
#define L_CMPX(r) "v_cmp_class_f32 vcc, " #r ", %8\n"
#define L_BPERM(r) "ds_bpermute_b32 " #r ", %8, " #r "\n s_waitcnt lgkmcnt(4)\n"
#define L_FMAMIX(r) "v_fma_f32 " #r ", " #r ", %8, %9\n v_max_f32 " #r ", " #r ", %9\n"
STREAM3(k_min, L_MIN)
STREAM3(k_max3, L_MAX3)
STREAM3(k_min3, L_MIN3)
STREAM3(k_sub, L_SUB)
STREAM3(k_and, L_AND)
STREAM3(k_xor, L_XOR)
STREAM3(k_lshl, L_LSHL)
STREAM3(k_addu, L_ADDU)
STREAM3(k_mad24, L_MAD24)
STREAM3(k_mul24, L_MUL24)
STREAM3(k_cvt, L_CVT)
STREAM3(k_bfe, L_BFE)
STREAM3(k_cmpcnd, L_CMPCND)
STREAM3(k_fmamax, L_FMAMIX)

// packed f32: 64-bit register pairs
typedef float f2 __attribute__((ext_vector_type(2)));
#define STREAM_PK(name, line)                                                                                      \
    __global__ __launch_bounds__(64) void name(Stamp* st, float* sink, int iters) {                                \
        f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f; \
        f2 b = {1.0000001f, 0.9999999f}, c = {0.5f, 0.25f};                                                        \
        asm volatile("" : "+v"(b), "+v"(c));                                                                       \
        const unsigned long long w0 = wall_clock64();                                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                \
        for (int i = 0; i < iters; i++)                                                                            \
            asm volatile(".rept 16\n" line(%0) line(%1) line(%2) line(%3) line(%4) line(%5) line(%6) line(%7) ".endr\n" \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        __builtin_amdgcn_s_waitcnt(0);                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                \
        const unsigned long long w1 = wall_clock64();                                                              \
        if (threadIdx.x == 0) {                                                                                    \
            Stamp s = {t0, t1, w0, w1, (uint32_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)),        \
                       (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)), {0, 0}};               \
            st[blockIdx.x] = s;                                                                                    \
        }                                                                                                          \
        const f2 r = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));                                            \
        sink[blockIdx.x * 64 + threadIdx.x] = r.x + r.y;                                                           \
    }
#define L_PKMUL(r) "v_pk_mul_f32 " #r ", " #r ", %8\n"
#define L_PKADD(r) "v_pk_add_f32 " #r ", " #r ", %9\n"
#define L_PKFMA(r) "v_pk_fma_f32 " #r ", " #r ", %8, %9\n"
STREAM_PK(k_pk_mul, L_PKMUL)
STREAM_PK(k_pk_add, L_PKADD)
STREAM_PK(k_pk_fma, L_PKFMA)

// a VALU stream with a scalar instruction after every vector instruction (the renderer issues 0.52 SALU per VALU):
// does the scalar unit's work share the SIMD's issue slots or ride along?
#define L_FMA_S(r) "v_fma_f32 " #r ", " #r ", %8, %9\n s_add_u32 s20, s20, 1\n"
#define STREAM_S(name, line)                                                                                       \
    __global__ __launch_bounds__(64) void name(Stamp* st, float* sink, int iters) {                                \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        float b = 1.0000001f, c = 0.5f;                                                                            \
        asm volatile("" : "+v"(b), "+v"(c));                                                                       \
        const unsigned long long w0 = wall_clock64();                                                              \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                \
        for (int i = 0; i < iters; i++)                                                                            \
            asm volatile(".rept 16\n" line(%0) line(%1) line(%2) line(%3) line(%4) line(%5) line(%6) line(%7) ".endr\n" \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s20", "s21", "scc", "vcc"); \
        __builtin_amdgcn_s_waitcnt(0);                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                \
        const unsigned long long w1 = wall_clock64();                                                              \
        if (threadIdx.x == 0) {                                                                                    \
            Stamp s = {t0, t1, w0, w1, (uint32_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)),        \
                       (uint32_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)), {0, 0}};               \
            st[blockIdx.x] = s;                                                                                    \
        }                                                                                                          \
        sink[blockIdx.x * 64 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));                   \
    }
STREAM_S(k_fma_salu, L_FMA_S)
STREAM_S(k_cnd64, L_CND64)
STREAM_S(k_cmps, L_CMPS)
STREAM_S(k_bperm, L_BPERM)

typedef void (*kern_t)(Stamp*, float*, int);
struct Case { const char* name; kern_t k; int valu_per_line; };

int main(int argc, char** argv) {
    int iters = argc > 1 ? atoi(argv[1]) : 4000;                 // 128 * 4000 = 512 000 wave-instructions per wave
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    const Case cases[] = {{"v_fma_f32", k_fma, 1}, {"v_mul_f32", k_mul, 1}, {"v_add_f32", k_add, 1}, {"v_max_f32", k_max, 1}, {"v_mov_b32", k_mov, 1},
                          {"v_cndmask_b32", k_cndmask, 1}, {"v_cmp_lt_f32", k_cmp, 1}, {"v_pk_mul_f32", k_pk_mul, 1}, {"v_pk_add_f32", k_pk_add, 1},
                          {"v_pk_fma_f32", k_pk_fma, 1}, {"v_mul_lo_u32", k_mullo, 1}, {"v_rcp_f32", k_rcp, 1}, {"v_sqrt_f32", k_sqrt, 1},
                          {"v_fma_f32 + s_add_u32", k_fma_salu, 1},
                          {"v_min_f32", k_min, 1}, {"v_max3_f32", k_max3, 1}, {"v_min3_f32", k_min3, 1}, {"v_sub_f32", k_sub, 1}, {"v_and_b32", k_and, 1},
                          {"v_xor_b32", k_xor, 1}, {"v_lshlrev_b32", k_lshl, 1}, {"v_add_u32", k_addu, 1}, {"v_mad_u32_u24", k_mad24, 1},
                          {"v_mul_u32_u24", k_mul24, 1}, {"v_cvt_f32_i32", k_cvt, 1}, {"v_bfe_u32", k_bfe, 1},
                          {"v_cndmask_b32_e64 (sgpr mask)", k_cnd64, 1}, {"v_cmp_lt_f32 -> sgpr pair", k_cmps, 1},
                          {"PAIR v_cmp_lt_f32 vcc + v_cndmask_b32 (per pair)", k_cmpcnd, 1}, {"PAIR v_fma_f32 + v_max_f32 (per pair)", k_fmamax, 1},
                          {"ds_bpermute_b32", k_bperm, 1}};
    const char* only = argc > 2 ? argv[2] : nullptr;                // "new": just the second batch
    const int waves_per_simd[] = {1, 2, 4, 5, 8};
    const size_t max_blocks = (size_t)simds * 8;
    Stamp* d_st; float* d_sink;
    CHECK(hipMalloc(&d_st, max_blocks * sizeof(Stamp)));
    CHECK(hipMalloc(&d_sink, max_blocks * 64 * sizeof(float)));
    std::vector<Stamp> h(max_blocks);
    printf("{\"device\": \"%s\", \"cus\": %d, \"simds\": %d, \"clock_rate_khz\": %d, \"instructions_per_wave\": %lld, \"results\": [\n", prop.gcnArchName, cus, simds,
           prop.clockRate, (long long)iters * PER_ITER);
    bool first = true;
    int case_index = 0;
    for (const Case& c : cases) {
        if (only && std::string(only) == "new" && case_index++ < 14) continue;
        // "calib": three streams at 1 and 5 waves per SIMD, to be run under rocprofv3 --pmc: what do SQ_ACTIVE_INST_VALU and
        // SQ_BUSY_CYCLES read for a stream whose issue rate is known?
        const bool calib = only && std::string(only) == "calib";
        if (calib && std::string(c.name) != "v_fma_f32" && std::string(c.name) != "v_max_f32" && std::string(c.name) != "v_pk_mul_f32") continue;
        for (int W : waves_per_simd) {
            if (calib && W != 1 && W != 5) continue;
            const int blocks = simds * W;
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, d_st, d_sink, 64);       // warm-up (code load, clocks)
            CHECK(hipDeviceSynchronize());
            hipEvent_t e0, e1;
            CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(64), 0, 0, d_st, d_sink, iters);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            CHECK(hipMemcpy(h.data(), d_st, blocks * sizeof(Stamp), hipMemcpyDeviceToHost));
            // group the waves by the SIMD they ran on
            struct Acc { int n = 0; unsigned long long lo = ~0ull, hi = 0; };
            std::map<uint64_t, Acc> by_simd;
            double clk_sum = 0;
            for (int b = 0; b < blocks; b++) {
                const Stamp& s = h[b];
                const uint64_t key = ((uint64_t)(s.xcc_id & 0xF) << 32) | (s.hw_id & 0x0000FF30u);   // xcc | se, sh, cu | simd
                Acc& a = by_simd[key];
                a.n++; a.lo = std::min(a.lo, s.t0); a.hi = std::max(a.hi, s.t1);
                clk_sum += (double)(s.t1 - s.t0) / ((double)(s.w1 - s.w0) / 100e6);                  // memtime ticks per second
            }
            const double per_wave_instr = (double)iters * PER_ITER;
            double cyc_sum = 0; int simds_exact = 0; int nmin = 1 << 30, nmax = 0;
            for (auto& kv : by_simd) {
                nmin = std::min(nmin, kv.second.n); nmax = std::max(nmax, kv.second.n);
                if (kv.second.n == W) { cyc_sum += (double)(kv.second.hi - kv.second.lo) / (per_wave_instr * W); simds_exact++; }
            }
            const double cyc = simds_exact ? cyc_sum / simds_exact : 0.0;
            const double clk_ghz = clk_sum / blocks / 1e9;
            // chip-wide view from the HIP events: wave-instructions per second per SIMD
            const double rate = per_wave_instr * blocks / (ms * 1e-3) / simds;
            printf("%s  {\"op\": \"%s\", \"waves_per_simd\": %d, \"simds_seen\": %zu, \"simds_with_exactly_W\": %d, \"waves_per_simd_min\": %d, \"waves_per_simd_max\": %d, "
                   "\"cycles_per_wave_instr_per_simd\": %.3f, \"memtime_ghz\": %.3f, \"kernel_ms\": %.3f, \"ginstr_per_s_per_simd_by_events\": %.4f, \"cycles_by_events_at_memtime_clock\": %.3f}",
                   first ? "" : ",\n", c.name, W, by_simd.size(), simds_exact, nmin, nmax, cyc, clk_ghz, ms, rate / 1e9, clk_ghz * 1e9 / rate);
            first = false;
            fflush(stdout);
            (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        }
    }
    printf("\n]}\n");
    return 0;
}
