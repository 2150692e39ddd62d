// cndmask / fma-bank follow-up probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ void __launch_bounds__(256) probe(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float s8 = 1.0000001f + seed * 1e-9f, s9 = 1e-9f + seed * 1e-12f, s10 = 0.5f + seed;
    unsigned long long mask = (threadIdx.x & 1) ? 0x5555555555555555ull : 0x3333333333333333ull;
    mask = __builtin_amdgcn_readfirstlane((unsigned)mask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mask >> 32)) << 32);
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { // VOP2 cndmask with vcc set once before
            asm volatile("s_mov_b64 vcc, %0" :: "s"(mask) : "vcc");
            REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                               "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10) : "vcc");)
        } else if (MODE == 1) { // VOP3 cndmask with an SGPR pair
            REP16(asm volatile("v_cndmask_b32 %0, %0, %8, %11\n v_cndmask_b32 %1, %1, %8, %11\n v_cndmask_b32 %2, %2, %8, %11\n v_cndmask_b32 %3, %3, %8, %11\n"
                               "v_cndmask_b32 %4, %4, %8, %11\n v_cndmask_b32 %5, %5, %8, %11\n v_cndmask_b32 %6, %6, %8, %11\n v_cndmask_b32 %7, %7, %8, %11\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10), "s"(mask));)
        } else if (MODE == 2) { // cndmask whose sources are two other registers (select, not self-update)
            REP16(asm volatile("v_cndmask_b32 %0, %9, %8, %11\n v_cndmask_b32 %1, %9, %8, %11\n v_cndmask_b32 %2, %9, %8, %11\n v_cndmask_b32 %3, %9, %8, %11\n"
                               "v_cndmask_b32 %4, %9, %8, %11\n v_cndmask_b32 %5, %9, %8, %11\n v_cndmask_b32 %6, %9, %8, %11\n v_cndmask_b32 %7, %9, %8, %11\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10), "s"(mask));)
        } else if (MODE == 3) { // cmp + cndmask pairs as the compiler emits for a ? b : c
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %10\n v_cndmask_b32 %1, %9, %8, vcc\n v_cmp_lt_f32 vcc, %2, %10\n v_cndmask_b32 %3, %9, %8, vcc\n"
                               "v_cmp_lt_f32 vcc, %4, %10\n v_cndmask_b32 %5, %9, %8, vcc\n v_cmp_lt_f32 vcc, %6, %10\n v_cndmask_b32 %7, %9, %8, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10) : "vcc");)
        } else if (MODE == 4) { // one cmp, then 7 cndmask (the TF pattern)
            REP16(asm volatile("v_cmp_lt_f32 vcc, %0, %10\n v_cndmask_b32 %1, %9, %8, vcc\n v_cndmask_b32 %2, %9, %8, vcc\n v_cndmask_b32 %3, %9, %8, vcc\n"
                               "v_cndmask_b32 %4, %9, %8, vcc\n v_cndmask_b32 %5, %9, %8, vcc\n v_cndmask_b32 %6, %9, %8, vcc\n v_cndmask_b32 %7, %9, %8, vcc\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10) : "vcc");)
        } else if (MODE == 5) { // fma with three distinct sources
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10));)
        } else if (MODE == 6) { // fma with a repeated source
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                               "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10));)
        } else if (MODE == 7) { // fmac (VOP2) with distinct sources
            REP16(asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n"
                               "v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10));)
        } else if (MODE == 8) { // fma writing a different register than it reads (a lerp: d = fma(w, e, s))
            REP16(asm volatile("v_fma_f32 %0, %1, %8, %9\n v_fma_f32 %1, %2, %8, %9\n v_fma_f32 %2, %3, %8, %9\n v_fma_f32 %3, %4, %8, %9\n"
                               "v_fma_f32 %4, %5, %8, %9\n v_fma_f32 %5, %6, %8, %9\n v_fma_f32 %6, %7, %8, %9\n v_fma_f32 %7, %0, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10));)
        } else if (MODE == 9) { // fma with an SGPR source
            REP16(asm volatile("v_fma_f32 %0, %0, %11, %9\n v_fma_f32 %1, %1, %11, %9\n v_fma_f32 %2, %2, %11, %9\n v_fma_f32 %3, %3, %11, %9\n"
                               "v_fma_f32 %4, %4, %11, %9\n v_fma_f32 %5, %5, %11, %9\n v_fma_f32 %6, %6, %11, %9\n v_fma_f32 %7, %7, %11, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10), "s"((unsigned)mask));)
        } else if (MODE == 10) { // v_sub_f32 on two denormal operands, then scale: the byte path
            REP16(asm volatile("v_sub_f32 %0, %11, %12\n v_mul_f32 %1, %8, %11\n v_sub_f32 %2, %11, %12\n v_mul_f32 %3, %8, %12\n"
                               "v_sub_f32 %4, %11, %12\n v_mul_f32 %5, %8, %11\n v_sub_f32 %6, %11, %12\n v_mul_f32 %7, %8, %12\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0x7f000000u), "v"(s9), "v"(s10), "v"(0x37u), "v"(0x91u));)
        } else if (MODE == 11) { // v_cmp_class / v_cmp_ge to an SGPR pair (VOP3)
            REP16(asm volatile("v_cmp_ge_f32 s[20:21], %0, %10\n v_cmp_ge_f32 s[22:23], %1, %10\n v_cmp_ge_f32 s[20:21], %2, %10\n v_cmp_ge_f32 s[22:23], %3, %10\n"
                               "v_cmp_ge_f32 s[20:21], %4, %10\n v_cmp_ge_f32 s[22:23], %5, %10\n v_cmp_ge_f32 s[20:21], %6, %10\n v_cmp_ge_f32 s[22:23], %7, %10\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s8), "v"(s9), "v"(s10) : "s20", "s21", "s22", "s23");)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void lds_unaligned_values(unsigned *out)
{
    __shared__ __attribute__((aligned(16))) unsigned char bytes[256];
    for (int i = threadIdx.x; i < 256; i += 64) bytes[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)bytes + threadIdx.x;
    unsigned h, w, b;
    asm volatile("ds_read_u16 %0, %3\n ds_read_b32 %1, %3\n ds_read_u8 %2, %3 offset:1\n s_waitcnt lgkmcnt(0)" : "=&v"(h), "=&v"(w), "=&v"(b) : "v"(a) : "memory");
    out[threadIdx.x] = h;
    out[64 + threadIdx.x] = w;
    out[128 + threadIdx.x] = b;
}
template <int M> void run(const char *name)
{
    float *out; (void)hipMalloc(&out, 1024 * 256 * 4);
    for (int w : {1, 4}) {
        probe<M><<<256 * w, 256>>>(out, 10, 1.0f); (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0); probe<M><<<256 * w, 256>>>(out, 1000, 1.0f); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-40s waves/SIMD=%d %.2f cycles per wave-inst\n", name, w, 2.4e9 / (1000.0 * 128 * w / (ms * 1e-3)));
    }
    (void)hipFree(out);
}
int main()
{
    run<0>("cndmask VOP2 vcc (self)"); run<1>("cndmask VOP3 sgpr (self)"); run<2>("cndmask VOP3 sgpr (select)");
    run<3>("cmp+cndmask pairs"); run<4>("1 cmp + 7 cndmask"); run<5>("fma distinct srcs"); run<6>("fma repeated src");
    run<7>("fmac distinct"); run<8>("fma rotating dst"); run<9>("fma sgpr src"); run<10>("sub/mul on denormals"); run<11>("v_cmp_ge -> sgpr pair");
    unsigned *d; (void)hipMalloc(&d, 192 * 4);
    lds_unaligned_values<<<1, 64>>>(d);
    unsigned h[192]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad16 = 0, bad32 = 0, bad8 = 0, bad16e = 0;
    auto b = [](int i) { return (unsigned)((i * 7 + 3) & 0xff); };
    for (int l = 0; l < 64; ++l) {
        if (h[l] != (b(l) | (b(l + 1) << 8))) { ++bad16; if (!(l & 1)) ++bad16e; if (bad16 < 6) printf("u16 @%d got %08x want %08x\n", l, h[l], b(l) | (b(l + 1) << 8)); }
        if (h[64 + l] != (b(l) | (b(l + 1) << 8) | (b(l + 2) << 16) | (b(l + 3) << 24))) ++bad32;
        if (h[128 + l] != b(l + 1)) ++bad8;
    }
    printf("u16 wrong %d (even addresses wrong %d), b32 wrong %d, u8 wrong %d of 64\n", bad16, bad16e, bad32, bad8);
    return 0;
}
