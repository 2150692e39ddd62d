// Probe 2 (round 3): issue cost of every VALU instruction class the march kernel's step loop and chunk prologue use or
// could use, and of the LDS byte / halfword reads an alpha tile addressed in bytes would need (with a check that a
// ds_read_u16 at an ODD byte address returns the two bytes it should on gfx950).  Same method as valu_probe.hip: 8
// independent chains x 16 per loop iteration, 1 / 4 waves per SIMD, cycles per wave-instruction per SIMD at 2.4 GHz
// from the event time.  Build: hipcc -O2 --offload-arch=gfx950 tools/probes/valu_probe2.hip -o tools/probes/bin/valu_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
// one group = the instruction on 8 independent destination registers; %8, %9 are two read-only sources
#define G8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)

#define PROBE(NAME, ASM8)                                                                                             \
    __global__ void __launch_bounds__(256) NAME(float *out, int iters, float seed)                                    \
    {                                                                                                                 \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        const float s8 = 1.0000001f + seed * 1e-9f;                                                                   \
        const unsigned s9 = __float_as_uint(seed) | 0x01020304u;                                                      \
        for (int i = 0; i < iters; ++i) {                                                                             \
            REP16(asm volatile(ASM8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)  \
                               : "v"(s8), "v"(s9) : "vcc");)                                                          \
        }                                                                                                             \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                           \
    }

#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %8\n"
#define I_FMAC(k) "v_fmac_f32 %" #k ", %8, %8\n"
#define I_ADDF(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %9\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %9\n"
#define I_LSHR(k) "v_lshrrev_b32 %" #k ", 3, %" #k "\n"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 2, %9\n"
#define I_MADU24(k) "v_mad_u32_u24 %" #k ", %" #k ", %9, %9\n"
#define I_MULU24(k) "v_mul_u32_u24 %" #k ", %" #k ", %9\n"
#define I_FLOOR(k) "v_floor_f32 %" #k ", %" #k "\n"
#define I_FRACT(k) "v_fract_f32 %" #k ", %" #k "\n"
#define I_CVTI(k) "v_cvt_i32_f32 %" #k ", %" #k "\n"
#define I_CVTFLR(k) "v_cvt_flr_i32_f32 %" #k ", %" #k "\n"
#define I_CVTF(k) "v_cvt_f32_i32 %" #k ", %" #k "\n"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %8\n"
#define I_MAX(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define I_CNDMASK(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %8\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_ALIGNBYTE(k) "v_alignbyte_b32 %" #k ", %" #k ", %9, %9\n"
#define I_CVTFP8SDWA(k) "v_cvt_f32_fp8_sdwa %" #k ", %9 src0_sel:BYTE_2\n"
#define I_DOT2C(k) "v_dot2c_f32_f16 %" #k ", %9, %9\n"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_LDEXP(k) "v_ldexp_f32 %" #k ", %" #k ", %9\n"
#define I_ADD3(k) "v_add3_u32 %" #k ", %" #k ", %9, %9\n"
#define I_MADI24(k) "v_mad_i32_i24 %" #k ", %" #k ", %9, %9\n"
#define I_FMA_LIT(k) "v_mul_f32 %" #k ", 0x3f800001, %" #k "\n"
#define I_SUB_SDWA(k) "v_sub_f32_sdwa %" #k ", %9, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_0\n"
#define I_MUL_SDWA_W(k) "v_mul_f32_sdwa %" #k ", %9, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define I_MINF(k) "v_min_f32 %" #k ", %" #k ", %8\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %9\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %9, 8, 8\n"
#define I_EXP(k) "v_exp_f32 %" #k ", %" #k "\n"
#define I_CVTUB(k) "v_cvt_f32_ubyte1 %" #k ", %9\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %9, %" #k ", %9\n"
#define I_SUBREV(k) "v_subrev_f32 %" #k ", %8, %" #k "\n"
#define I_XOR(k) "v_xor_b32 %" #k ", %" #k ", %9\n"
#define I_READLANE(k) "v_readlane_b32 s20, %" #k ", 5\n"

PROBE(p_fma, G8(I_FMA))
PROBE(p_fmac, G8(I_FMAC))
PROBE(p_addf, G8(I_ADDF))
PROBE(p_addu, G8(I_ADDU))
PROBE(p_and, G8(I_AND))
PROBE(p_lshr, G8(I_LSHR))
PROBE(p_lshladd, G8(I_LSHLADD))
PROBE(p_madu24, G8(I_MADU24))
PROBE(p_mulu24, G8(I_MULU24))
PROBE(p_floor, G8(I_FLOOR))
PROBE(p_fract, G8(I_FRACT))
PROBE(p_cvti, G8(I_CVTI))
PROBE(p_cvtflr, G8(I_CVTFLR))
PROBE(p_cvtf, G8(I_CVTF))
PROBE(p_med3, G8(I_MED3))
PROBE(p_max, G8(I_MAX))
PROBE(p_cndmask, G8(I_CNDMASK))
PROBE(p_cmp, G8(I_CMP))
PROBE(p_mov, G8(I_MOV))
PROBE(p_alignbyte, G8(I_ALIGNBYTE))
PROBE(p_cvtfp8sdwa, G8(I_CVTFP8SDWA))
PROBE(p_dot2c, G8(I_DOT2C))
PROBE(p_rcp, G8(I_RCP))
PROBE(p_ldexp, G8(I_LDEXP))
PROBE(p_add3, G8(I_ADD3))
PROBE(p_madi24, G8(I_MADI24))
PROBE(p_fma_lit, G8(I_FMA_LIT))
PROBE(p_sub_sdwa, G8(I_SUB_SDWA))
PROBE(p_mul_sdwa_w, G8(I_MUL_SDWA_W))
PROBE(p_minf, G8(I_MINF))
PROBE(p_mullo, G8(I_MULLO))
PROBE(p_bfe, G8(I_BFE))
PROBE(p_exp, G8(I_EXP))
PROBE(p_cvtub, G8(I_CVTUB))
PROBE(p_perm, G8(I_PERM))
PROBE(p_xor, G8(I_XOR))

// packed: two results per instruction (register pairs)
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) probe_pk(float *out, int iters, float seed)
{
    float2v p0 = {seed, seed + 1}, p1 = p0 + 1.0f, p2 = p0 + 2.0f, p3 = p0 + 3.0f, p4 = p0 + 4.0f, p5 = p0 + 5.0f, p6 = p0 + 6.0f, p7 = p0 + 7.0f;
    const float2v m2 = {1.0000001f, 1.0000001f};
    const unsigned w = __float_as_uint(seed) | 0x38404448u; // fp8 bytes
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                               "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(w));)
        } else if (MODE == 1) {
            REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                               "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(w));)
        } else if (MODE == 2) {
            REP16(asm volatile("v_cvt_pk_f32_fp8 %0, %9\n v_cvt_pk_f32_fp8 %1, %9\n v_cvt_pk_f32_fp8 %2, %9\n v_cvt_pk_f32_fp8 %3, %9\n"
                               "v_cvt_pk_f32_fp8 %4, %9\n v_cvt_pk_f32_fp8 %5, %9\n v_cvt_pk_f32_fp8 %6, %9\n v_cvt_pk_f32_fp8 %7, %9\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(w));)
        } else if (MODE == 3) {
            REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                               "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8\n"
                               : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(m2), "v"(w));)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
}

// v_readlane_b32 into SGPRs (the prologue's spilled-SGPR traffic)
__global__ void __launch_bounds__(256) p_readlane(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x;
    unsigned acc = 0;
    for (int i = 0; i < iters; ++i) {
        REP16(asm volatile("v_readlane_b32 s20, %1, 1\n v_readlane_b32 s21, %1, 2\n v_readlane_b32 s22, %1, 3\n v_readlane_b32 s23, %1, 4\n"
                           "v_readlane_b32 s24, %1, 5\n v_readlane_b32 s25, %1, 6\n v_readlane_b32 s26, %1, 7\n v_readlane_b32 s27, %1, 8\n"
                           "s_add_u32 %0, %0, s27\n"
                           : "+s"(acc) : "v"(a0) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + (float)acc;
}

// ---- LDS read probes: one wave per workgroup x 4 per SIMD, a 10 KB tile each (the march kernel's shape) -----------------
// MODE 0: 4 x ds_read2_b32 offset1:1 (today's gather), 1: 8 x ds_read_u8, 2: 4 x ds_read_u16 (odd addresses allowed),
// 3: 4 x ds_read_b32, 4: 4 x ds_read_u16 at even addresses.  Addresses: a pseudo-random walk like the march's gathers.
template <int MODE>
__global__ void __launch_bounds__(64) probe_lds(unsigned *out, int iters, unsigned seed)
{
    __shared__ __attribute__((aligned(16))) unsigned tile[2560];
    for (int i = threadIdx.x; i < 2560; i += 64) tile[i] = i * 0x01010101u + seed;
    __syncthreads();
    unsigned acc = 0;
    // lane (x, y) of an 8x8 tile, box pitch 48 bytes, slice 49*... : a0 walks through the tile by a lane-dependent stride
    const unsigned lx = threadIdx.x & 7u, ly = threadIdx.x >> 3;
    unsigned pos = (lx * 5u + ly * 3u * 48u + seed) % 8000u;
    const unsigned pitch = 48u, slice = 48u * 12u + 16u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll 8
        for (int r = 0; r < 8; ++r) {
            unsigned a0 = pos;
            if (MODE == 0 || MODE == 3) a0 &= ~3u;
            if (MODE == 4) a0 &= ~1u;
            const unsigned a1 = a0 + pitch, a2 = a0 + slice, a3 = a2 + pitch;
            unsigned v0, v1, v2, v3, v4, v5, v6, v7;
            if (MODE == 0) {
                unsigned long long q0, q1, q2, q3;
                asm volatile("ds_read2_b32 %0, %4 offset1:1\n ds_read2_b32 %1, %5 offset1:1\n ds_read2_b32 %2, %6 offset1:1\n ds_read2_b32 %3, %7 offset1:1\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
                acc += (unsigned)q0 + (unsigned)(q0 >> 32) + (unsigned)q1 + (unsigned)(q1 >> 32) + (unsigned)q2 + (unsigned)(q2 >> 32) + (unsigned)q3 + (unsigned)(q3 >> 32);
            } else if (MODE == 1) {
                asm volatile("ds_read_u8 %0, %8\n ds_read_u8 %1, %8 offset:1\n ds_read_u8 %2, %9\n ds_read_u8 %3, %9 offset:1\n"
                             "ds_read_u8 %4, %10\n ds_read_u8 %5, %10 offset:1\n ds_read_u8 %6, %11\n ds_read_u8 %7, %11 offset:1\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3), "=&v"(v4), "=&v"(v5), "=&v"(v6), "=&v"(v7)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
                acc += v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
            } else if (MODE == 2 || MODE == 4) {
                asm volatile("ds_read_u16 %0, %4\n ds_read_u16 %1, %5\n ds_read_u16 %2, %6\n ds_read_u16 %3, %7\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
                acc += v0 + v1 + v2 + v3;
            } else {
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %5\n ds_read_b32 %2, %6\n ds_read_b32 %3, %7\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
                acc += v0 + v1 + v2 + v3;
            }
            pos += 25u + lx; // half a texel or so per step, lane-dependent
            if (pos >= 8000u) pos -= 8000u;
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

// value check: ds_read_u16 / ds_read_b32 at every byte offset 0..7 of a known pattern
__global__ void lds_unaligned_values(unsigned *out)
{
    __shared__ __attribute__((aligned(16))) unsigned char bytes[256];
    for (int i = threadIdx.x; i < 256; i += 64) bytes[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const unsigned a = threadIdx.x; // byte address = lane
    unsigned h, w;
    asm volatile("ds_read_u16 %0, %2\n ds_read_b32 %1, %2\n s_waitcnt lgkmcnt(0)" : "=&v"(h), "=&v"(w) : "v"(a) : "memory");
    out[threadIdx.x] = h;
    out[64 + threadIdx.x] = w;
}

static double time_ms(void (*launch)(int iters), int iters)
{
    launch(10);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    launch(iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

static float *g_out;
static int g_blocks;
#define RUN(NAME, LABEL)                                                                                   \
    for (int w : {1, 4}) {                                                                                 \
        g_blocks = 256 * w;                                                                                \
        const int iters = 1000;                                                                            \
        const double ms = time_ms([](int it) { NAME<<<g_blocks, 256>>>(g_out, it, 1.0f); }, iters);        \
        const double per_simd = (double)iters * 128 * w / (ms * 1e-3);                                     \
        printf("%-28s waves/SIMD=%d  %.2f cycles per wave-inst per SIMD at 2.4 GHz\n", LABEL, w, 2.4e9 / per_simd); \
    }

int main()
{
    (void)hipMalloc(&g_out, (size_t)1024 * 256 * 4 * 2);
    RUN(p_fma, "v_fma_f32")
    RUN(p_fmac, "v_fmac_f32")
    RUN(p_fma_lit, "v_mul_f32 literal")
    RUN(p_addf, "v_add_f32")
    RUN(p_minf, "v_min_f32")
    RUN(p_max, "v_max_f32")
    RUN(p_med3, "v_med3_f32")
    RUN(p_floor, "v_floor_f32")
    RUN(p_fract, "v_fract_f32")
    RUN(p_cvti, "v_cvt_i32_f32")
    RUN(p_cvtflr, "v_cvt_flr_i32_f32")
    RUN(p_cvtf, "v_cvt_f32_i32")
    RUN(p_cvtub, "v_cvt_f32_ubyte1")
    RUN(p_ldexp, "v_ldexp_f32")
    RUN(p_rcp, "v_rcp_f32")
    RUN(p_exp, "v_exp_f32")
    RUN(p_mov, "v_mov_b32")
    RUN(p_cndmask, "v_cndmask_b32")
    RUN(p_cmp, "v_cmp_lt_f32")
    RUN(p_addu, "v_add_u32")
    RUN(p_and, "v_and_b32")
    RUN(p_xor, "v_xor_b32")
    RUN(p_lshr, "v_lshrrev_b32")
    RUN(p_lshladd, "v_lshl_add_u32")
    RUN(p_add3, "v_add3_u32")
    RUN(p_madu24, "v_mad_u32_u24")
    RUN(p_madi24, "v_mad_i32_i24")
    RUN(p_mulu24, "v_mul_u32_u24")
    RUN(p_mullo, "v_mul_lo_u32")
    RUN(p_bfe, "v_bfe_u32")
    RUN(p_perm, "v_perm_b32")
    RUN(p_alignbyte, "v_alignbyte_b32")
    RUN(p_sub_sdwa, "v_sub_f32_sdwa(byte,byte)")
    RUN(p_mul_sdwa_w, "v_mul_f32_sdwa(word)")
    RUN(p_cvtfp8sdwa, "v_cvt_f32_fp8_sdwa")
    RUN(p_dot2c, "v_dot2c_f32_f16")
    RUN(p_readlane, "v_readlane_b32 (+1/8 s_add)")
    for (int w : {1, 4}) {
        g_blocks = 256 * w;
        const char *names[4] = {"v_pk_mul_f32", "v_pk_add_f32", "v_cvt_pk_f32_fp8", "v_pk_fma_f32"};
        double ms[4];
        ms[0] = time_ms([](int it) { probe_pk<0><<<g_blocks, 256>>>(g_out, it, 1.0f); }, 1000);
        ms[1] = time_ms([](int it) { probe_pk<1><<<g_blocks, 256>>>(g_out, it, 1.0f); }, 1000);
        ms[2] = time_ms([](int it) { probe_pk<2><<<g_blocks, 256>>>(g_out, it, 1.0f); }, 1000);
        ms[3] = time_ms([](int it) { probe_pk<3><<<g_blocks, 256>>>(g_out, it, 1.0f); }, 1000);
        for (int k = 0; k < 4; ++k)
            printf("%-28s waves/SIMD=%d  %.2f cycles per wave-inst per SIMD at 2.4 GHz\n", names[k], w, 2.4e9 / (1000.0 * 128 * w / (ms[k] * 1e-3)));
    }
    // LDS: 4096 one-wave workgroups (16 per CU), cycles per CU per "sample" (one group of reads by one wave)
    {
        const char *names[5] = {"4 x ds_read2_b32", "8 x ds_read_u8", "4 x ds_read_u16 (any addr)", "4 x ds_read_b32", "4 x ds_read_u16 (even addr)"};
        double ms[5];
        g_blocks = 4096;
        ms[0] = time_ms([](int it) { probe_lds<0><<<g_blocks, 64>>>((unsigned *)g_out, it, 1u); }, 2000);
        ms[1] = time_ms([](int it) { probe_lds<1><<<g_blocks, 64>>>((unsigned *)g_out, it, 1u); }, 2000);
        ms[2] = time_ms([](int it) { probe_lds<2><<<g_blocks, 64>>>((unsigned *)g_out, it, 1u); }, 2000);
        ms[3] = time_ms([](int it) { probe_lds<3><<<g_blocks, 64>>>((unsigned *)g_out, it, 1u); }, 2000);
        ms[4] = time_ms([](int it) { probe_lds<4><<<g_blocks, 64>>>((unsigned *)g_out, it, 1u); }, 2000);
        for (int k = 0; k < 5; ++k) {
            const double samples_per_cu = 2000.0 * 8 * 16; // per CU: 16 waves x iters x 8
            printf("LDS %-28s %.1f CU-cycles per wave-sample at 2.4 GHz (16 waves per CU, includes the address adds)\n", names[k],
                   ms[k] * 1e-3 * 2.4e9 / samples_per_cu);
        }
    }
    {
        unsigned *d;
        (void)hipMalloc(&d, 128 * 4);
        lds_unaligned_values<<<1, 64>>>(d);
        unsigned h[128];
        (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad16 = 0, bad32 = 0;
        for (int l = 0; l < 64; ++l) {
            auto b = [](int i) { return (unsigned)((i * 7 + 3) & 0xff); };
            const unsigned want16 = b(l) | (b(l + 1) << 8);
            const unsigned want32 = b(l) | (b(l + 1) << 8) | (b(l + 2) << 16) | (b(l + 3) << 24);
            if (h[l] != want16) { if (bad16 < 4) printf("u16 at byte %d: got %08x want %08x\n", l, h[l], want16); ++bad16; }
            if (h[64 + l] != want32) { if (bad32 < 4) printf("b32 at byte %d: got %08x want %08x\n", l, h[64 + l], want32); ++bad32; }
        }
        printf("unaligned LDS reads: ds_read_u16 %s (%d of 64 wrong), ds_read_b32 %s (%d of 64 wrong)\n", bad16 ? "NOT byte-addressed" : "exact at every byte address", bad16,
               bad32 ? "NOT byte-addressed" : "exact at every byte address", bad32);
    }
    return 0;
}
