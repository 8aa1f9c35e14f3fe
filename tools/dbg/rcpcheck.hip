// Exhaustive check: is fma(fma(-a, r, 1), r, r) with r = v_rcp_f32(a) equal to the correctly rounded 1.0f / a for every
// binary32 a with lo <= |a| < hi?  Default range 1e-4 .. 2^100 (rcp_det); `rcpcheck -100 100` checks 2^-100 .. 2^100 (rcp_exact).   (build: hipcc --offload-arch=gfx950 -O2 -ffp-contract=off rcpcheck.hip -o rcpcheck)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <cmath>
__global__ void k(unsigned long long* bad, uint32_t* ex, uint32_t lo, uint32_t hi) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)lo + blockIdx.x * blockDim.x + threadIdx.x; b < hi; b += stride) {
        for (int sgn = 0; sgn < 2; sgn++) {
            const float a = __uint_as_float((uint32_t)b | (sgn ? 0x80000000u : 0u));
            const float want = 1.0f / a;
            const float r = __builtin_amdgcn_rcpf(a);
            const float e = fmaf(-a, r, 1.0f);
            const float got = fmaf(e, r, r);
            if (__float_as_uint(got) != __float_as_uint(want)) {
                const unsigned long long n = atomicAdd(bad, 1ull);
                if (n < 16) ex[n] = __float_as_uint(a);
            }
        }
    }
}
int main(int argc, char** argv) {
    unsigned long long* bad; uint32_t* ex;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&ex, 64); (void)hipMemset(bad, 0, 8); (void)hipMemset(ex, 0, 64);
    float lof = 1.0e-4f, hif = 1.2676506e30f;   // 2^100
    if (argc > 2) { lof = ldexpf(1.0f, atoi(argv[1])); hif = ldexpf(1.0f, atoi(argv[2])); }
    uint32_t lo, hi; std::memcpy(&lo, &lof, 4); std::memcpy(&hi, &hif, 4);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, bad, ex, lo, hi);
    unsigned long long hb = 0; uint32_t hex[16];
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hex, ex, 64, hipMemcpyDeviceToHost);
    printf("range %.9g .. %.9g: checked %u bit patterns x 2 signs, mismatches: %llu\n", lof, hif, hi - lo, hb);
    for (int i = 0; i < 16 && i < (int)hb; i++) { float f; std::memcpy(&f, &hex[i], 4); printf("  a = %.9g (0x%08x)\n", f, hex[i]); }
    return 0;
}
