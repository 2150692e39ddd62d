// Probe: which device spellings of f32 sqrt / divide are correctly rounded on gfx950 (vs host IEEE)?
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#pragma clang fp contract(off)
__global__ void probe(const float *a, const float *b, float *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    o[0 * n + i] = sqrtf(a[i]);
    o[1 * n + i] = __fsqrt_rn(a[i]);
    o[2 * n + i] = __builtin_sqrtf(a[i]);
    o[3 * n + i] = a[i] / b[i];
    o[4 * n + i] = __fdiv_rn(a[i], b[i]);
    o[5 * n + i] = a[i] * b[i] + a[i];      // must NOT be fused
    o[6 * n + i] = (a[i] * a[i] + b[i] * b[i]) + a[i] * b[i];
}
int main()
{
    const int n = 1 << 20;
    std::vector<float> a(n), b(n), o(7 * n);
    srand(1);
    for (int i = 0; i < n; ++i) { a[i] = 0.001f + 4.0f * rand() / RAND_MAX; b[i] = 0.001f + 3.0f * rand() / RAND_MAX; }
    float *da, *db, *dout;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dout, 7 * n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<n / 256, 256>>>(da, db, dout, n);
    hipMemcpy(o.data(), dout, 7 * n * 4, hipMemcpyDeviceToHost);
    const char *names[7] = {"sqrtf", "__fsqrt_rn", "__builtin_sqrtf", "a/b", "__fdiv_rn", "a*b+a unfused", "sumsq"};
    for (int k = 0; k < 7; ++k) {
        int bad = 0;
        for (int i = 0; i < n; ++i) {
            volatile float m = a[i] * b[i];
            volatile float s1 = a[i] * a[i], s2 = b[i] * b[i];
            volatile float s12 = s1 + s2;
            float ref = k < 3 ? sqrtf(a[i]) : k < 5 ? a[i] / b[i] : k == 5 ? m + a[i] : s12 + m;
            bad += (ref != o[k * n + i]);
        }
        printf("%-18s mismatches vs host: %d / %d\n", names[k], bad, n);
    }
    return 0;
}
