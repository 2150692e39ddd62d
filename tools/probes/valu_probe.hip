// Probe: VALU issue rates on gfx950 that decide how the trilinear filter is written:
// v_fma_f32 vs v_pk_fma_f32 (two f32 FMAs per lane per instruction), v_cvt_f32_ubyteN, v_exp_f32,
// at 1, 2 and 4 waves per SIMD.  Prints wave-instructions per SIMD-cycle.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ void __launch_bounds__(256) probe(float *out, unsigned long long *cycles, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 1.0000001f, c = 1e-9f;
    const float2v m2 = {m, m}, c2 = {c, c};
    unsigned int w = __float_as_uint(seed) | 0x01020304u;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (MODE == 1) {
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                               "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(c2));)
        } else if (MODE == 2) {
            REP16(asm volatile("v_cvt_f32_ubyte0 %0, %8\n v_cvt_f32_ubyte1 %1, %8\n v_cvt_f32_ubyte2 %2, %8\n v_cvt_f32_ubyte3 %3, %8\n"
                               "v_cvt_f32_ubyte0 %4, %8\n v_cvt_f32_ubyte1 %5, %8\n v_cvt_f32_ubyte2 %6, %8\n v_cvt_f32_ubyte3 %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));)
        } else if (MODE == 3) {
            REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                               "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (MODE == 4) { // fma_mix with one f16 operand (low half of %8)
            REP16(asm volatile("v_fma_mix_f32 %0, %8, %9, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %8, %9, %1 op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %2, %8, %9, %2 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %8, %9, %3 op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %4, %8, %9, %4 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %5, %8, %9, %5 op_sel_hi:[1,0,0]\n"
                               "v_fma_mix_f32 %6, %8, %9, %6 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %7, %8, %9, %7 op_sel_hi:[1,0,0]\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(c));)
        } else if (MODE == 5) { // byte -> biased float through one byte permute (selector in %9)
            REP16(asm volatile("v_perm_b32 %0, %8, %0, %9\n v_perm_b32 %1, %8, %1, %9\n v_perm_b32 %2, %8, %2, %9\n v_perm_b32 %3, %8, %3, %9\n"
                               "v_perm_b32 %4, %8, %4, %9\n v_perm_b32 %5, %8, %5, %9\n v_perm_b32 %6, %8, %6, %9\n v_perm_b32 %7, %8, %7, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(0x07060004u));)
        } else if (MODE == 6) { // packed f16 FMA (two halves per lane)
            REP16(asm volatile("v_pk_fma_f16 %0, %0, %8, %9\n v_pk_fma_f16 %1, %1, %8, %9\n v_pk_fma_f16 %2, %2, %8, %9\n v_pk_fma_f16 %3, %3, %8, %9\n"
                               "v_pk_fma_f16 %4, %4, %8, %9\n v_pk_fma_f16 %5, %5, %8, %9\n v_pk_fma_f16 %6, %6, %8, %9\n v_pk_fma_f16 %7, %7, %8, %9\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(c));)
        } else if (MODE == 7) { // bit-field extract (the integer alternative to cvt_ubyte)
            REP16(asm volatile("v_bfe_u32 %0, %8, 8, 8\n v_bfe_u32 %1, %8, 16, 8\n v_bfe_u32 %2, %8, 8, 8\n v_bfe_u32 %3, %8, 16, 8\n"
                               "v_bfe_u32 %4, %8, 8, 8\n v_bfe_u32 %5, %8, 16, 8\n v_bfe_u32 %6, %8, 8, 8\n v_bfe_u32 %7, %8, 16, 8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w));)
        } else if (MODE == 8) { // plain f32 add (VOP2)
            REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                               "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));)
        } else if (MODE == 9) { // cvt_ubyte through SDWA on a VOP2 add: a + float? no -- u8 dot: v_dot4_u32_u8
            REP16(asm volatile("v_dot4_u32_u8 %0, %8, %9, %0\n v_dot4_u32_u8 %1, %8, %9, %1\n v_dot4_u32_u8 %2, %8, %9, %2\n v_dot4_u32_u8 %3, %8, %9, %3\n"
                               "v_dot4_u32_u8 %4, %8, %9, %4\n v_dot4_u32_u8 %5, %8, %9, %5\n v_dot4_u32_u8 %6, %8, %9, %6\n v_dot4_u32_u8 %7, %8, %9, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(0x01010101u));)
        } else if (MODE == 10) { // byte (as a denormal bit pattern) * 2^127 through SDWA: convert and scale in one VOP2
            REP16(asm volatile("v_mul_f32_sdwa %0, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                               "v_mul_f32_sdwa %1, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                               "v_mul_f32_sdwa %2, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
                               "v_mul_f32_sdwa %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
                               "v_mul_f32_sdwa %4, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
                               "v_mul_f32_sdwa %5, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
                               "v_mul_f32_sdwa %6, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD\n"
                               "v_mul_f32_sdwa %7, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(0x7f000000u));)
        } else if (MODE == 11) { // byte - byte in the denormal domain
            REP16(asm volatile("v_sub_f32_sdwa %0, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n"
                               "v_sub_f32_sdwa %1, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                               "v_sub_f32_sdwa %2, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2\n"
                               "v_sub_f32_sdwa %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n"
                               "v_sub_f32_sdwa %4, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0\n"
                               "v_sub_f32_sdwa %5, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1\n"
                               "v_sub_f32_sdwa %6, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2\n"
                               "v_sub_f32_sdwa %7, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(0x04030201u));)
        } else if (MODE == 12) { // FMA with a denormal multiplicand (the difference above) and a 2^127-scaled weight
            REP16(asm volatile("v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n"
                               "v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(0x00000037u), "v"(0x7e800000u));)
        } else if (MODE == 13) { // the other half-rate suspects of the inner loop
            REP16(asm volatile("v_floor_f32 %0, %0\n v_floor_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n"
                               "v_med3_f32 %4, %4, %8, %9\n v_med3_f32 %5, %5, %8, %9\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(c));)
        } else if (MODE == 14) { // f32 multiply (VOP2)
            REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                               "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.x;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// out[l] = byte1(w) * 2^-22, out[64+l] = (byte3 - byte1) * 2^-24, out[128+l] = their sum through one FMA
__global__ void sdwa_values(float *out)
{
    const unsigned l = threadIdx.x;
    const unsigned w = (((l * 101u + 7u) & 0xffu) << 24) | (((l * 37u + 11u) & 0xffu) << 8) | 0x00aa0055u;
    float m, d, f;
    const unsigned two127 = 0x7f000000u, two125 = 0x7e000000u;
    asm volatile("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(m) : "v"(w), "v"(two127));
    asm volatile("v_sub_f32_sdwa %0, %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_1" : "=v"(d) : "v"(w));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(f) : "v"(d), "v"(two125), "v"(m));
    float dm;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(dm) : "v"(d), "v"(two125));
    out[l] = m;
    out[64 + l] = dm;
    out[128 + l] = f;
}

template <int MODE>
void run(const char *name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd; // 256-thread blocks = 4 waves = one wave per SIMD of a CU
    const int iters = 2000;
    float *out;
    unsigned long long *cyc;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    (void)hipMalloc(&cyc, (size_t)blocks * 8);
    probe<MODE><<<blocks, 256>>>(out, cyc, 10, 1.0f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<MODE><<<blocks, 256>>>(out, cyc, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks;
    const double insts = (double)iters * 16 * 8;          // per wave
    // s_memtime ticks at 100 MHz on gfx9 (constant clock); convert with the event time instead
    const double wave_inst_per_s = insts * waves_per_simd * 1024.0 / (ms * 1e-3);
    printf("%-22s waves/SIMD=%d  time %.3f ms  => %.2f G wave-inst/s/SIMD  (%.2f cycles per wave-inst per SIMD at 2.4 GHz), memtime ticks/wave %.0f\n",
           name, waves_per_simd, ms, wave_inst_per_s / 1024.0 / 1e9, 2.4e9 / (wave_inst_per_s / 1024.0), mean);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", w);
        run<1>("v_pk_fma_f32", w);
        run<2>("v_cvt_f32_ubyteN", w);
        run<3>("v_exp_f32", w);
        run<4>("v_fma_mix_f32(f16 src)", w);
        run<5>("v_perm_b32", w);
        run<6>("v_pk_fma_f16", w);
        run<7>("v_bfe_u32", w);
        run<8>("v_add_f32", w);
        run<9>("v_dot4_u32_u8", w);
        run<10>("v_mul_f32_sdwa(byte)", w);
        run<11>("v_sub_f32_sdwa(bytes)", w);
        run<12>("v_fma_f32(denormal src)", w);
        run<13>("floor/cvt_i32/med3/add_u32", w);
        run<14>("v_mul_f32", w);
    }
    // value check of the denormal-domain byte arithmetic
    {
        float *d;
        (void)hipMalloc(&d, 64 * 3 * sizeof(float));
        sdwa_values<<<1, 64>>>(d);
        float h[192];
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l) {
            const unsigned b1 = (l * 37u + 11u) & 0xffu, b3 = (l * 101u + 7u) & 0xffu;
            const float want_mul = (float)b1 * 0x1p-22f;                       // byte1 * 2^-149 * 2^127
            const float want_sub = ((float)b3 - (float)b1) * 0x1p-22f * 0.25f; // (byte3 - byte1) * 2^-149 * (0.25 * 2^127)
            const float want_fma = want_sub + want_mul;
            if (h[l] != want_mul || h[64 + l] != want_sub || h[128 + l] != want_fma) {
                if (bad < 4) printf("lane %d: mul %g (want %g) sub %g (want %g) fma %g (want %g)\n", l, h[l], want_mul, h[64 + l], want_sub, h[128 + l], want_fma);
                ++bad;
            }
        }
        printf("sdwa denormal-domain values: %s (%d mismatches of 64)\n", bad ? "MISMATCH" : "exact", bad);
    }
    return 0;
}
