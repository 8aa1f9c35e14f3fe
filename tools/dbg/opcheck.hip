// Checks that device f32 divide / sqrt / uint->float are correctly rounded (vs host IEEE).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
__global__ void k(const float* a, const float* b, float* q, float* r, float* s, float* m, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q[i] = a[i] / b[i];
    r[i] = 1.0f / b[i];
    s[i] = sqrtf(fabsf(a[i]));
    m[i] = a[i] * (1.0f / sqrtf((a[i] * a[i] + b[i] * b[i]) + 1.0f));
}
int main() {
    const int n = 1 << 22;
    std::vector<float> a(n), b(n), q(n), r(n), s(n), m(n);
    uint32_t st = 12345;
    for (int i = 0; i < n; i++) {
        st = st * 1664525u + 1013904223u; uint32_t x = st;
        st = st * 1664525u + 1013904223u; uint32_t y = st;
        float fa, fb;
        if (i & 1) { fa = (float)(x >> 8) * 5.9604645e-8f * 20.f - 10.f; fb = (float)(y >> 8) * 5.9604645e-8f * 20.f - 10.f; }
        else { x = (x & 0x807FFFFF) | (((x >> 23) % 60 + 97) << 23); y = (y & 0x807FFFFF) | (((y >> 23) % 60 + 97) << 23); memcpy(&fa, &x, 4); memcpy(&fb, &y, 4); }
        a[i] = fa; b[i] = fb;
    }
    float *da, *db, *dq, *dr, *ds, *dm;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dq, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dm, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(da, db, dq, dr, ds, dm, n);
    hipMemcpy(q.data(), dq, n * 4, hipMemcpyDeviceToHost); hipMemcpy(r.data(), dr, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost);
    long bq = 0, br = 0, bs = 0, bm = 0;
    for (int i = 0; i < n; i++) {
        volatile float hq = a[i] / b[i], hr = 1.0f / b[i], hs = sqrtf(fabsf(a[i]));
        volatile float t0 = a[i] * a[i]; volatile float t1 = b[i] * b[i]; volatile float t2 = t0 + t1; volatile float t3 = t2 + 1.0f;
        volatile float t4 = sqrtf(t3); volatile float t5 = 1.0f / t4; volatile float hm = a[i] * t5;
        float x;
        x = hq; bq += memcmp(&x, &q[i], 4) != 0; x = hr; br += memcmp(&x, &r[i], 4) != 0; x = hs; bs += memcmp(&x, &s[i], 4) != 0;
        x = hm; if (memcmp(&x, &m[i], 4) != 0) { if (bm < 5) printf("norm mismatch a=%a b=%a host=%a dev=%a\n", a[i], b[i], hm, m[i]); bm++; }
    }
    printf("mismatches of %d: div %ld rcp %ld sqrt %ld normalize-expr %ld\n", n, bq, br, bs, bm);
    return 0;
}
