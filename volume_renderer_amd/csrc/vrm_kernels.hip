// vrm_kernels.hip -- hand-written gfx950 (CDNA4 / MI355X) kernels of the ray-marching path.
//
// What is computed is the reference's fragment shader (src/ray_marcher/raymarcher.wgsl:98-299);
// how it is computed is this file's own design (DESIGN.md "Kernels").  The parity contract splits
// the arithmetic in two:
//   * geometry (ray direction, slab test, the running dt sum, the sample position and the in-box
//     test) feeds discontinuous decisions, so it is evaluated with exactly the oracle's sequence of
//     separately rounded f32 operations -- hence contraction is OFF for the whole file and fused
//     operations are written explicitly;
//   * filter / transfer function / blend are continuous, written with explicit FMAs and folded
//     constants, and agree with the oracle to rounding.
#include <type_traits>

#include "vrm_device.h"

#pragma clang fp contract(off)

#ifndef VRM_CAP_DEFAULT
#ifndef VRM_PK_ASM
#define VRM_PK_ASM 1 /* the RGBA filter's packed lerps written out, the weights in ONE register pair (0: left to the compiler; A/B) */
#endif
#define VRM_CAP_DEFAULT 2560 /* LDS words per wave of the default march (A/B builds override it) */
#endif
#ifndef VRM_CAP_PACKED
#define VRM_CAP_PACKED VRM_CAP_DEFAULT /* LDS words per wave of the packed 1-channel march */
#endif
#ifndef VRM_INTERIOR_NO_CLAMP
#define VRM_INTERIOR_NO_CLAMP 1 /* A/B: 0 keeps the LDS address clamp in interior chunks too */
#endif
#ifndef VRM_WEIGHTS_FIRST
#define VRM_WEIGHTS_FIRST 0 /* A/B: trilinear filter of the RGBA march as eight corner weights + byte * weight products */
#endif
#ifndef VRM_RAYS2_DEPTH
#define VRM_RAYS2_DEPTH 15.0f /* texels of depth per chunk of the two-rays-per-lane march */
#endif
#ifndef VRM_SKIP_MAX_LEVEL
#define VRM_SKIP_MAX_LEVEL 3 /* deepest occupancy level used for empty-space strides (0 = plain chunk-by-chunk skipping) */
#endif
#ifndef VRM_XLERP_PRODUCTS
#define VRM_XLERP_PRODUCTS 1 /* x interpolation of the byte-tile filter as two products instead of difference + product (A/B: 0) */
#endif
#ifndef VRM_NO_SLP_TAIL
#define VRM_NO_SLP_TAIL 1 /* blend tail of the byte-tile filter as five plain instructions, kept out of v_pk_* pairs that cost three v_mov (A/B: 0) */
#endif
#ifndef VRM_TRANSMITTANCE
/* A/B, measured and NOT adopted: kernels without the early exit carry the transmittance T = 1 - alpha (T *= e, weight = T - T*e:
 * two instructions fewer per sample, -1 % on C3, -1.5 % on C4) -- but the weight then carries the rounding of T*e, not of
 * 1 - e, and the f32 accumulators drift from the oracle's twice as far (fuzz seed 41, 1268 checks: 99.9 % 5.6e-5 -> 1.07e-4,
 * max 7.0e-5 -> 1.26e-4, four cases beyond the 1e-4 contract; profiles/r3/transmittance.txt) */
#define VRM_TRANSMITTANCE 0
#endif
#ifndef VRM_SKIP_START_LEVEL
#define VRM_SKIP_START_LEVEL 2 /* the level a tile's first chunk is probed at: rays enter through empty space more often than not (A/B: 0) */
#endif
#ifndef VRM_SKIP_FALLBACK
#define VRM_SKIP_FALLBACK 2 /* a failed deep probe drops ONE level and waits this many more empty chunks before it probes deeper again (A/B: 0 = back to level 0) */
#endif
#ifndef VRM_SKIP_ESCALATE
#define VRM_SKIP_ESCALATE 1  /* consecutive empty chunks at one level before the next level is tried */
#endif

namespace {

constexpr int TILE = 16;        // variant 0: workgroup = 16x16 pixels = 4 waves, each wave an 8x8 pixel tile
constexpr float LOG2E = 1.44269504088896340736f;

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  Give each XCD a
// contiguous range of image tiles so neighbouring tiles -- which read neighbouring voxels -- share
// an L2.  Bijective for any grid size (cdna_hip_programming.md T1).
__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t n)
{
    const uint32_t xcd = b & 7u, slot = b >> 3;
    const uint32_t q = n >> 3, r = n & 7u;
    const uint32_t base = (xcd < r) ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q;
    return base + slot;
}

struct Ray {
    float ro[3], rd[3];
    float t_min, t_max;
    bool hit;
};

// fs_main (wgsl:98-108) + aabb_intersect (wgsl:247-299); op-for-op the oracle's shade_pixel().
// sqrtf and '/' are the correctly rounded forms (-fhip-fp32-correctly-rounded-divide-sqrt); the
// __fsqrt_rn intrinsic is NOT correctly rounded on gfx950 (tools/probes/fp_probe.hip).
__device__ __forceinline__ Ray setup_ray(const VrmViewDesc &v, const VrmGridDev &g, uint32_t px, uint32_t py, uint32_t W,
                                         uint32_t H)
{
    Ray r;
    const float uvx = ((float)px + 0.5f) / (float)W;
    const float uvy = 1.0f - ((float)py + 0.5f) / (float)H;
    const float sx = uvx * 2.0f - 1.0f, sy = uvy * 2.0f - 1.0f;
    float w3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) w3[k] = ((v.ivp[0 + k] * sx + v.ivp[4 + k] * sy) + v.ivp[8 + k] * 1.0f) + v.ivp[12 + k] * 1.0f;
    const float len = __builtin_sqrtf((w3[0] * w3[0] + w3[1] * w3[1]) + w3[2] * w3[2]);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.rd[k] = w3[k] / len;
        r.ro[k] = v.pos[k];
    }
    float t_min = (g.box_min[0] - r.ro[0]) / r.rd[0];
    float t_max = (g.box_max[0] - r.ro[0]) / r.rd[0];
    if (t_min > t_max) { float t = t_min; t_min = t_max; t_max = t; }
    float t_y_min = (g.box_min[1] - r.ro[1]) / r.rd[1];
    float t_y_max = (g.box_max[1] - r.ro[1]) / r.rd[1];
    if (t_y_min > t_y_max) { float t = t_y_min; t_y_min = t_y_max; t_y_max = t; }
    bool hit = !((t_min > t_y_max) || (t_y_min > t_max));
    if (t_y_min > t_min) t_min = t_y_min;
    if (t_y_max < t_max) t_max = t_y_max;
    float t_z_min = (g.box_min[2] - r.ro[2]) / r.rd[2];
    float t_z_max = (g.box_max[2] - r.ro[2]) / r.rd[2];
    if (t_z_min > t_z_max) { float t = t_z_min; t_z_min = t_z_max; t_z_max = t; }
    hit = hit && !((t_min > t_z_max) || (t_z_min > t_max));
    if (t_z_min > t_min) t_min = t_z_min;
    if (t_z_max < t_max) t_max = t_z_max;
    r.t_min = t_min;
    r.t_max = t_max;
    r.hit = hit;
    return r;
}

__device__ __forceinline__ uint32_t unorm8(float v)
{
    // Rgba8Unorm store: round(clamp(v,0,1)*255); NaN -> 0 (fmaxf(NaN,0) = 0)
    v = fminf(fmaxf(v, 0.0f), 1.0f);
    return (uint32_t)floorf(v * 255.0f + 0.5f);
}

constexpr float T22 = 4194304.0f; // 2^22 = 2^149 / 2^127

// byte K of w, as a float in units of 2^-22: the byte read through SDWA is the denormal K * 2^-149, times 2^127
template <int K>
__device__ __forceinline__ float byte_scaled(uint32_t w)
{
    static_assert(K >= 0 && K < 4, "byte index");
    float r;
    const float two127 = 0x1p127f;
    if constexpr (K == 0) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(two127));
    if constexpr (K == 1) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(two127));
    if constexpr (K == 2) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(two127));
    if constexpr (K == 3) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(two127));
    return r;
}

// (byte K of b) - (byte K of a) as the exact denormal difference * 2^-149 (multiply by a 2^127-scaled weight to use it)
template <int K>
__device__ __forceinline__ float byte_diff(uint32_t b, uint32_t a)
{
    float r;
    if constexpr (K == 0) asm("v_sub_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(r) : "v"(b), "v"(a));
    if constexpr (K == 1) asm("v_sub_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(r) : "v"(b), "v"(a));
    if constexpr (K == 2) asm("v_sub_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(r) : "v"(b), "v"(a));
    if constexpr (K == 3) asm("v_sub_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(r) : "v"(b), "v"(a));
    return r;
}

struct Accum {
    float cr, cg, cb, alpha; // colour in 0..255 texel units while marching when the TF is off
};

// The transfer function's three colours, held in VGPRs on purpose: the march kernel is short of SGPRs, and nine uniform
// floats that every sample reads were what the register allocator spilled to VGPR lanes and read back per sample
// (v_readlane + v_mov in the step loop).  Held in the select-free form of tf_blend() below: colour A and the doubled
// differences 2(B - A), 2(C - B).
struct TfRegs {
    float a[3], d1[3], d2[3];
};
template <bool USE_TF>
__device__ __forceinline__ TfRegs tf_regs(const VrmMarchParams &P)
{
    TfRegs t = {};
    if (USE_TF) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            t.a[i] = P.col_a[i];
            t.d1[i] = (P.col_b[i] - P.col_a[i]) * 2.0f;
            t.d2[i] = (P.col_c[i] - P.col_b[i]) * 2.0f;
            asm volatile("" : "+v"(t.a[i]), "+v"(t.d1[i]), "+v"(t.d2[i]));
        }
    }
    return t;
}

// The transfer function (wgsl:145-147: A -> B for a in [0, 0.5], B -> C above, piecewise linear) times the sample's weight,
// added to the colour.  Written without selects: with h = max(a - 0.5, 0) and l = a - h = min(a, 0.5) (both exact)
// colour = A + l * 2(B - A) + h * 2(C - B).  On gfx950 a v_cndmask_b32 that takes its mask from VCC costs 20+ cycles
// unless it directly follows the compare (tools/probes/valu_probe3.hip: one compare + seven selects, the form the compiler
// gives `hi ? x : y` on nine colours, averages 16 cycles per instruction against 2.5-4.3 for everything else); this form
// is one v_max and eight full-rate operations.  a22: the filtered alpha in units of 2^-22 texel levels.
__device__ __forceinline__ void tf_blend(const float a22, const float w, const TfRegs &P, Accum &acc)
{
    const float a = a22 * (T22 * (1.0f / 255.0f));
    const float h = fmaxf(a - 0.5f, 0.0f);
    const float l = a - h;
    acc.cr = __builtin_fmaf(w, __builtin_fmaf(h, P.d2[0], __builtin_fmaf(l, P.d1[0], P.a[0])), acc.cr);
    acc.cg = __builtin_fmaf(w, __builtin_fmaf(h, P.d2[1], __builtin_fmaf(l, P.d1[1], P.a[1])), acc.cg);
    acc.cb = __builtin_fmaf(w, __builtin_fmaf(h, P.d2[2], __builtin_fmaf(l, P.d1[2], P.a[2])), acc.cb);
}

// Trilinear filter of the 8 corner texels (c[z][y][x] as RGBA8 words) + transfer function (wgsl:145-147)
// + Beer-Lambert opacity (wgsl:150) + front-to-back blend (wgsl:158-159).  Returns alpha >= 1 (wgsl:176).
// Filter arithmetic runs in units of 2^-22 texel levels (see byte_scaled): `a22` below is the filtered alpha in those units,
// kexp22 = kexp * 2^22 / 255... folded by the caller so that a22 * kexp22 = -a * step * attenuation * log2(e).
template <bool ALPHA_ONLY, bool USE_TF, bool ERT>
__device__ __forceinline__ bool shade(const uint32_t c[8], float wx, float wy, float wz, float kexp22, const TfRegs &P,
                                      Accum &acc, float *a_out = nullptr)
{
    // Byte k of an RGBA8 word read through SDWA is the f32 denormal k * 2^-149: one instruction gives the difference
    // of two texels' channels, exact, and an FMA with the 2^127-scaled weight brings it back into the normal range
    // (tools/probes/valu_probe.hip: values exact, the FMA runs at full rate with a denormal operand).  Per x-lerp
    // that is sub_sdwa + mul_sdwa + fma instead of two conversions, a subtraction and an FMA.
    const float wxs = wx * 0x1p127f;
#if VRM_WEIGHTS_FIRST
    // A/B form: the eight corner weights once per sample (shared by the channels), then per channel eight
    // byte * weight products -- the SDWA multiply converts and scales in one instruction -- and an add tree.  gfx950 has no
    // multiply-accumulate with a byte-selected operand (v_mac/v_fmac_f32_sdwa do not assemble), so the accumulate is a
    // separate add.
    float W[8];
    {
        const float w11 = wx * wy, w10 = wx - w11, w01 = wy - w11, w00 = (1.0f - wx) - w01; // (x,y) = (1,1), (1,0), (0,1), (0,0)
        const float z1 = wz * 0x1p127f;
        W[4] = w00 * z1; W[5] = w10 * z1; W[6] = w01 * z1; W[7] = w11 * z1;
        W[0] = __builtin_fmaf(w00, 0x1p127f, -W[4]); W[1] = __builtin_fmaf(w10, 0x1p127f, -W[5]);
        W[2] = __builtin_fmaf(w01, 0x1p127f, -W[6]); W[3] = __builtin_fmaf(w11, 0x1p127f, -W[7]);
    }
    auto tri = [&](auto kc) -> float {
        constexpr int k = decltype(kc)::value;
        auto bw = [&](uint32_t t, float w) -> float {
            float r;
            if constexpr (k == 0) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(t), "v"(w));
            if constexpr (k == 1) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(t), "v"(w));
            if constexpr (k == 2) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(t), "v"(w));
            if constexpr (k == 3) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(t), "v"(w));
            return r;
        };
        return ((bw(c[0], W[0]) + bw(c[1], W[1])) + (bw(c[2], W[2]) + bw(c[3], W[3]))) +
               ((bw(c[4], W[4]) + bw(c[5], W[5])) + (bw(c[6], W[6]) + bw(c[7], W[7])));
    };
#else
    auto tri = [&](auto kc) -> float {
        constexpr int k = decltype(kc)::value;
        const float a00 = byte_scaled<k>(c[0]), a10 = byte_scaled<k>(c[2]), a01 = byte_scaled<k>(c[4]), a11 = byte_scaled<k>(c[6]);
        const float c00 = __builtin_fmaf(byte_diff<k>(c[1], c[0]), wxs, a00);
        const float c10 = __builtin_fmaf(byte_diff<k>(c[3], c[2]), wxs, a10);
        const float c01 = __builtin_fmaf(byte_diff<k>(c[5], c[4]), wxs, a01);
        const float c11 = __builtin_fmaf(byte_diff<k>(c[7], c[6]), wxs, a11);
        const float c0 = __builtin_fmaf(wy, c10 - c00, c00);
        const float c1 = __builtin_fmaf(wy, c11 - c01, c01);
        return __builtin_fmaf(wz, c1 - c0, c0);
    };
#endif
#if VRM_PK_ASM && !VRM_WEIGHTS_FIRST
    // The same lerps as tri(), in the same operations and order (bit-identical: tools/ab_bits.py), written as the packed
    // instructions the compiler's SLP pass pairs them into, but with the weights in ONE register pair {wx * 2^127, wy} picked per
    // instruction by op_sel: the compiler builds a {w, w} pair per weight instead, two v_mov per sample (84 -> 82 VALU
    // instructions in the step body, 123 -> 121 VGPRs; C3 -1.9 %, profiles/r3/pk_asm.txt).  Pairs: one channel of two rows (y0z0
    // with y0z1, y1z0 with y1z1: the y-lerp then runs in both halves and leaves {value at z = 0, value at z = 1}), or red and
    // green of one row.  One statement per x-y stage: the compiler pads every inline-assembly result that the next instruction
    // reads with an s_nop (17 per sample as single-instruction statements, 5 so; dependent v_pk_*_f32 need none on gfx950).
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 wq = {wxs, wy};
    f2 wzq;
    wzq.x = wz; // (the high half is never read: op_sel_hi = 0 on this operand)
    // lo = fma(dlo, wxs, alo), hi = fma(dhi, wxs, ahi), then fma(wy, hi - lo, lo), each in both halves
    auto xyl = [&](f2 dlo, f2 alo, f2 dhi, f2 ahi) -> f2 {
        f2 lo, t, r;
        asm("v_pk_fma_f32 %0, %3, %7, %4 op_sel_hi:[1,0,1]\n\t"
            "v_pk_fma_f32 %1, %5, %7, %6 op_sel_hi:[1,0,1]\n\t"
            "v_pk_add_f32 %1, %1, %0 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_fma_f32 %2, %7, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
            : "=&v"(lo), "=&v"(t), "=&v"(r) : "v"(dlo), "v"(alo), "v"(dhi), "v"(ahi), "v"(wq));
        return r;
    };
    auto rows_a = [&](auto kc, int i, int j) -> f2 { constexpr int k = decltype(kc)::value; return f2{byte_scaled<k>(c[2 * i]), byte_scaled<k>(c[2 * j])}; };
    auto rows_d = [&](auto kc, int i, int j) -> f2 {
        constexpr int k = decltype(kc)::value;
        return f2{byte_diff<k>(c[2 * i + 1], c[2 * i]), byte_diff<k>(c[2 * j + 1], c[2 * j])};
    };
    auto rg_a = [&](int i) -> f2 { return f2{byte_scaled<0>(c[2 * i]), byte_scaled<1>(c[2 * i])}; };
    auto rg_d = [&](int i) -> f2 { return f2{byte_diff<0>(c[2 * i + 1], c[2 * i]), byte_diff<1>(c[2 * i + 1], c[2 * i])}; };
    // channel k of rows (i, j) against rows (i + 1, j + 1): x-lerps, then the y-lerp in both halves
    auto rows_xy = [&](auto kc, int i, int j) -> f2 {
        return xyl(rows_d(kc, i, j), rows_a(kc, i, j), rows_d(kc, i + 1, j + 1), rows_a(kc, i + 1, j + 1));
    };
    // red and green of rows i (y = 0) and i + 1 (y = 1)
    auto rg_xy = [&](int i) -> f2 { return xyl(rg_d(i), rg_a(i), rg_d(i + 1), rg_a(i + 1)); };
    const f2 a01 = rows_xy(std::integral_constant<int, 3>{}, 0, 2); // rows y0z0, y0z1 against y1z0, y1z1
    const float a22 = __builtin_fmaf(wz, a01.y - a01.x, a01.x);
    (void)rg_xy;
    (void)tri;
#else
    const float a22 = tri(std::integral_constant<int, 3>{});
#endif
    if (a_out) *a_out = a22;
    // alpha_src = 1 - exp(-a*step*att) = 1 - exp2(a22 * kexp22)
    const float e = __builtin_amdgcn_exp2f(a22 * kexp22);
    const float alpha_src = 1.0f - e;
    // (TRANSMITTANCE: acc.alpha holds T = 1 - alpha; T' = T * e and the sample's weight (1 - alpha) * alpha_src = T - T')
    constexpr bool TR = !ERT && VRM_TRANSMITTANCE != 0;
    const float om = TR ? acc.alpha : 1.0f - acc.alpha;
    const float t1 = om * e;
    const float w = TR ? om - t1 : om * alpha_src;
    if (USE_TF) {
        tf_blend(a22, w, P, acc);
    } else if (ALPHA_ONLY) {
        acc.cr = __builtin_fmaf(w, a22, acc.cr); // r == g == b == a for scalar volumes (finish_colour() rescales)
    } else if (__builtin_amdgcn_ballot_w64(a22 != 0.0f) != 0ull) {
        // a == 0 => alpha_src = 0 => w = 0 => the colour terms add exactly 0: filter r,g,b only when some
        // lane of the wave has a non-zero alpha (wave-uniform branch, three quarters of the filter work)
#if VRM_PK_ASM && !VRM_WEIGHTS_FIRST
        // red and green of a row as one pair, blue of two rows as one pair (see the helpers above)
        const f2 rg0 = rg_xy(0), rg1 = rg_xy(2); // z = 0, z = 1
        f2 t, rgz;
        asm("v_pk_add_f32 %0, %2, %3 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_fma_f32 %1, %4, %0, %3 op_sel_hi:[0,1,1]" : "=&v"(t), "=&v"(rgz) : "v"(rg1), "v"(rg0), "v"(wzq));
        const f2 b01 = rows_xy(std::integral_constant<int, 2>{}, 0, 2);
        const float bz = __builtin_fmaf(wz, b01.y - b01.x, b01.x);
        acc.cr = __builtin_fmaf(w, rgz.x, acc.cr);
        acc.cg = __builtin_fmaf(w, rgz.y, acc.cg);
        acc.cb = __builtin_fmaf(w, bz, acc.cb);
#else
        acc.cr = __builtin_fmaf(w, tri(std::integral_constant<int, 0>{}), acc.cr);
        acc.cg = __builtin_fmaf(w, tri(std::integral_constant<int, 1>{}), acc.cg);
        acc.cb = __builtin_fmaf(w, tri(std::integral_constant<int, 2>{}), acc.cb);
#endif
    }
    acc.alpha = TR ? t1 : __builtin_fmaf(om, alpha_src, acc.alpha);
    // wgsl:176.  Without ERT the launcher has established that no sample can reach alpha_src >= 0.5 (attenuation < 690), and
    // then alpha never rounds up to 1.0: the test is dead and left out.
    return ERT ? acc.alpha >= 1.0f : false;
}

// The 1-channel march from the packed alpha copy (VrmGridDev::abricks), whose LDS tile is addressed in BYTES: t[2r] and
// t[2r + 1] are the alpha bytes of texels x and x + 1 of row r (rows: y0z0, y1z0, y0z1, y1z1), each read by its own
// ds_read_u8.  A zero-extended byte IS the f32 denormal b * 2^-149, so the filter is plain full-rate f32 arithmetic with
// no conversion at all: the difference of two denormals is exact, a multiply by 2^127 and an FMA with the 2^127-scaled
// weight bring them into the normal range.  (8 ds_read_u8 cost the LDS what 4 ds_read2_b32 did, tools/probes/valu_probe2.hip.)
// Every sample of a PACKED instance goes through this function, the global-gather fallback chunks included: the x
// interpolation rounds differently from shade<true, USE_TF>'s (two products against difference + product), and a sample's
// contribution must not depend on how the march was cut into chunks.
template <bool USE_TF, bool ERT>
__device__ __forceinline__ bool shade_bytes(const uint32_t t[8], float wx, float wy, float wz, float kexp22, const TfRegs &P,
                                            Accum &acc, float *a_out = nullptr)
{
    const float wxs = wx * 0x1p127f;
#if VRM_XLERP_PRODUCTS
    // b0 * (1 - wx) + b1 * wx with both weights scaled by 2^127: a multiply and an FMA per row instead of a subtraction, a
    // multiply and an FMA (-3 instructions per sample; continuous arithmetic, another rounding of the same value)
    const float w0s = __builtin_fmaf(-wx, 0x1p127f, 0x1p127f);
    auto xlerp = [&](uint32_t t0, uint32_t t1) -> float {
        const float b0 = __builtin_bit_cast(float, t0), b1 = __builtin_bit_cast(float, t1);
        return __builtin_fmaf(b1, wxs, b0 * w0s);
    };
#else
    auto xlerp = [&](uint32_t t0, uint32_t t1) -> float {
        const float b0 = __builtin_bit_cast(float, t0), b1 = __builtin_bit_cast(float, t1);
        return __builtin_fmaf(b1 - b0, wxs, b0 * 0x1p127f);
    };
#endif
    const float c00 = xlerp(t[0], t[1]), c10 = xlerp(t[2], t[3]), c01 = xlerp(t[4], t[5]), c11 = xlerp(t[6], t[7]);
    const float c0 = __builtin_fmaf(wy, c10 - c00, c00);
    const float c1 = __builtin_fmaf(wy, c11 - c01, c01);
    const float a22 = __builtin_fmaf(wz, c1 - c0, c0);
    if (a_out) *a_out = a22;
    const float e = __builtin_amdgcn_exp2f(a22 * kexp22);
    const float alpha_src = 1.0f - e;
    constexpr bool TR = !ERT && VRM_TRANSMITTANCE != 0; // (see shade())
    float om = TR ? acc.alpha : 1.0f - acc.alpha;
#if VRM_NO_SLP_TAIL
    // (opaque to the SLP vectoriser, which pairs 1 - alpha with 1 - e and the two accumulations into v_pk_* instructions and
    // then spends three v_mov on packing their operands: five plain full-rate instructions instead)
    asm("" : "+v"(om));
#endif
    const float t1 = om * e;
    const float wgt = TR ? om - t1 : om * alpha_src;
    if (USE_TF) tf_blend(a22, wgt, P, acc);
#if VRM_NO_SLP_TAIL
    else asm("v_fmac_f32 %0, %1, %2" : "+v"(acc.cr) : "v"(wgt), "v"(a22)); // (= the fmaf below, kept out of a v_pk_fma_f32 with the alpha update)
#else
    else acc.cr = __builtin_fmaf(wgt, a22, acc.cr); // r == g == b == a for scalar volumes (finish_colour() rescales)
#endif
    acc.alpha = TR ? t1 : __builtin_fmaf(om, alpha_src, acc.alpha);
    return ERT ? acc.alpha >= 1.0f : false;
}

template <bool ALPHA_ONLY, bool USE_TF, bool ERT>
__device__ __forceinline__ void finish_colour(Accum &acc)
{
    if (!ERT && VRM_TRANSMITTANCE != 0) acc.alpha = 1.0f - acc.alpha; // the march carried the transmittance (see shade())
    if (!USE_TF) {
        const float s = T22 * (1.0f / 255.0f); // colours were accumulated in units of 2^-22 texel levels
        acc.cr *= s;
        if (ALPHA_ONLY) {
            acc.cg = acc.cr;
            acc.cb = acc.cr;
        } else {
            acc.cg *= s;
            acc.cb *= s;
        }
    }
}

template <bool DIAG>
__device__ __forceinline__ void store_pixel(const VrmViewDesc &v, const VrmMarchParams &P, uint32_t px, uint32_t py,
                                            const Accum &acc, const Ray &ray, uint32_t iters, uint32_t n_inside, uint32_t n_exec)
{
    const uint32_t packed = unorm8(acc.cr) | (unorm8(acc.cg) << 8) | (unorm8(acc.cb) << 16) | (unorm8(acc.alpha) << 24);
    if (v.out) v.out[(size_t)py * P.pitch_px + px] = packed;
    if (v.partial_rgba) // wave-uniform
        reinterpret_cast<float4 *>(v.partial_rgba)[(size_t)py * P.width + px] = make_float4(acc.cr, acc.cg, acc.cb, acc.alpha);
    if (DIAG) {
        const size_t o = (size_t)py * P.width + px;
        if (v.diag_rgba) reinterpret_cast<float4 *>(v.diag_rgba)[o] = make_float4(acc.cr, acc.cg, acc.cb, acc.alpha);
        // low half: loop iterations (<= 5000); high half: samples actually filtered (N_exec) -- fewer than the
        // in-box count when empty chunks were skipped
        if (v.diag_iters) v.diag_iters[o] = iters | (n_exec << 16);
        if (v.diag_geom) {
            float4 *gq = reinterpret_cast<float4 *>(v.diag_geom) + 2 * o;
            gq[0] = make_float4(ray.rd[0], ray.rd[1], ray.rd[2], ray.hit ? ray.t_min : 0.0f);
            gq[1] = make_float4(ray.hit ? ray.t_max : 0.0f, (float)n_inside, ray.hit ? 1.0f : 0.0f, (float)iters);
        }
    }
}

// 8 corner texels straight from the x-fastest global volume, clamp-to-edge (texture_3d.rs:66-76)
__device__ __forceinline__ void fetch_global(const VrmGridDev &g, int ix, int iy, int iz, uint32_t c[8])
{
    const uint32_t X = g.dim[0], Y = g.dim[1];
    const int mx = (int)g.dim[0] - 1, my = (int)g.dim[1] - 1, mz = (int)g.dim[2] - 1;
    const uint32_t x0 = (uint32_t)min(max(ix, 0), mx), x1 = (uint32_t)max(min(ix + 1, mx), 0);
    const uint32_t y0 = (uint32_t)min(max(iy, 0), my), y1 = (uint32_t)max(min(iy + 1, my), 0);
    const uint32_t z0 = (uint32_t)min(max(iz, 0), mz), z1 = (uint32_t)max(min(iz + 1, mz), 0);
    const size_t r00 = ((size_t)z0 * Y + y0) * X, r10 = ((size_t)z0 * Y + y1) * X;
    const size_t r01 = ((size_t)z1 * Y + y0) * X, r11 = ((size_t)z1 * Y + y1) * X;
    c[0] = g.vox[r00 + x0]; c[1] = g.vox[r00 + x1];
    c[2] = g.vox[r10 + x0]; c[3] = g.vox[r10 + x1];
    c[4] = g.vox[r01 + x0]; c[5] = g.vox[r01 + x1];
    c[6] = g.vox[r11 + x0]; c[7] = g.vox[r11 + x1];
}

// The packed alpha bricks by texel (VrmGridDev::abricks; x, y, z inside the volume): word index of the four texels
// 4*xw .. 4*xw+3 of row (y, z), and one texel's alpha.  What the packed march reads where it cannot stage whole quads, so
// that a scalar volume needs neither the x-fastest RGBA copy nor the RGBA bricks to be resident.
__device__ __forceinline__ size_t abrick_word(const VrmGridDev &g, uint32_t xw, uint32_t y, uint32_t z)
{
    return ((((size_t)(z >> 1) * g.nby + (y >> 2)) * g.nbxa + (xw >> 2)) << 5) + ((z & 1u) << 4) + ((y & 3u) << 2) + (xw & 3u);
}
__device__ __forceinline__ uint32_t abrick_alpha(const VrmGridDev &g, uint32_t x, uint32_t y, uint32_t z)
{
    return (g.abricks[abrick_word(g, x >> 2, y, z)] >> (8u * (x & 3u))) & 0xffu;
}
// 8 corner texels of a scalar volume from the packed alpha bricks, as words with the alpha byte set (clamp-to-edge)
__device__ __forceinline__ void fetch_global_alpha(const VrmGridDev &g, int ix, int iy, int iz, uint32_t c[8])
{
    const int mx = (int)g.dim[0] - 1, my = (int)g.dim[1] - 1, mz = (int)g.dim[2] - 1;
    const uint32_t x0 = (uint32_t)min(max(ix, 0), mx), x1 = (uint32_t)max(min(ix + 1, mx), 0);
    const uint32_t y0 = (uint32_t)min(max(iy, 0), my), y1 = (uint32_t)max(min(iy + 1, my), 0);
    const uint32_t z0 = (uint32_t)min(max(iz, 0), mz), z1 = (uint32_t)max(min(iz + 1, mz), 0);
    c[0] = abrick_alpha(g, x0, y0, z0) << 24; c[1] = abrick_alpha(g, x1, y0, z0) << 24;
    c[2] = abrick_alpha(g, x0, y1, z0) << 24; c[3] = abrick_alpha(g, x1, y1, z0) << 24;
    c[4] = abrick_alpha(g, x0, y0, z1) << 24; c[5] = abrick_alpha(g, x1, y0, z1) << 24;
    c[6] = abrick_alpha(g, x0, y1, z1) << 24; c[7] = abrick_alpha(g, x1, y1, z1) << 24;
}

// ---------------------------------------------------------------------------------------------
// Variant 0: software trilinear straight from global memory.  One thread per pixel, one wave per
// 8x8 pixel tile, 4 waves per workgroup.  Correctness baseline and A/B reference for variant 1.
// ---------------------------------------------------------------------------------------------
template <bool ALPHA_ONLY, bool USE_TF, bool DIAG>
__global__ void __launch_bounds__(256) march_linear_kernel(const VrmMarchParams P)
{
    const VrmGridDev &g = P.grid;
    const uint32_t view = blockIdx.y;
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t px = tx * TILE + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t py = ty * TILE + (wave >> 1) * 8u + (lane >> 3);
    if (px >= P.width || py >= P.height) return;
    const VrmViewDesc &v = P.views[view];
    const Ray ray = setup_ray(v, g, px, py, P.width, P.height);

    Accum acc = {0.0f, 0.0f, 0.0f, 0.0f};
    uint32_t iters = 0, n_inside = 0;
    if (ray.hit) {
        const float kexp = -(VRM_STEP_SIZE * P.attenuation) * LOG2E * (1.0f / 255.0f) * T22;
        const TfRegs tf = tf_regs<USE_TF>(P);
        float dt = ray.t_min;
        for (int i = 0; i < VRM_MAX_STEP_AMOUNT; ++i) {
            ++iters;
            // exact geometry: p = ro + rd*dt, separately rounded (wgsl:141)
            const float p0 = ray.ro[0] + ray.rd[0] * dt;
            const float p1 = ray.ro[1] + ray.rd[1] * dt;
            const float p2 = ray.ro[2] + ray.rd[2] * dt;
            const bool inside = (p0 >= g.box_min[0]) & (p1 >= g.box_min[1]) & (p2 >= g.in_min_z) & (p0 < g.box_max[0]) &
                                (p1 < g.box_max[1]) & (p2 < g.in_max_z); // wgsl:218-223
            if (inside) {
                if (DIAG) ++n_inside;
                const float u0 = __builtin_fmaf(p0, g.tex_scale[0], g.tex_bias[0]);
                const float u1 = __builtin_fmaf(p1, g.tex_scale[1], g.tex_bias[1]);
                const float u2 = __builtin_fmaf(p2, g.tex_scale[2], g.tex_bias[2]);
                const float f0 = floorf(u0), f1 = floorf(u1), f2 = floorf(u2);
                uint32_t c[8];
                fetch_global(g, (int)f0, (int)f1, (int)f2, c);
                if (shade<ALPHA_ONLY, USE_TF, true>(c, u0 - f0, u1 - f1, u2 - f2, kexp, tf, acc)) break; // wgsl:176-178
            }
            dt += VRM_STEP_SIZE;          // wgsl:181, running f32 sum
            if (dt >= ray.t_max) break;   // wgsl:184-186
        }
        finish_colour<ALPHA_ONLY, USE_TF, true>(acc);
    }
    store_pixel<DIAG>(v, P, px, py, acc, ray, iters, n_inside, n_inside);
}

// ---------------------------------------------------------------------------------------------
// Variant 1: per-wave LDS staging.
//
// One wave (= one 64-thread workgroup) owns an 8x8 pixel tile and marches its 64 rays in chunks.  Per chunk each lane
// takes the number of steps (0..K) that brings it up to a plane across the wave's major axis; the wave computes the
// axis-aligned box (in texel space) that holds every texel those steps can touch, skips the chunk if the occupancy grid
// says the box is empty, otherwise copies the box from the bricked volume into its LDS tile by LDS-DMA and takes all
// 8 gathers per sample from LDS.  A box that does not fit the tile (far-away cameras: one pixel spans many voxels)
// falls back to global gathers for that chunk.  The dt lattice, the sample positions and the in-box test are the same
// exact sequence as in variant 0 / the oracle; only where the texels are read from differs.
// ---------------------------------------------------------------------------------------------
// Wave64 min/max on the VALU's data-parallel-primitive path (no LDS traffic, unlike __shfl): two quad
// permutes, two row mirrors, then row_bcast:15 / row_bcast:31 carry the row results upwards; lane 63 ends up
// with the reduction over all 64 lanes and is read back into an SGPR.  Must run with all lanes enabled
// (inactive rays pass the identity), which is how the kernel calls it.
// The compiler turns update_dpp + fminf into mov, nop, mov_dpp, canonicalise, min per stage (26 VALU instructions per
// reduction, and the chunk prologue does eight of them); written out, a stage is ONE v_min_f32_dpp whose first source is
// the permuted copy of the register itself.  A DPP read of a VGPR needs two wait states after the VALU write of it:
// s_nop 1 between dependent stages, or the other chain's stage plus s_nop 0 when a min and a max run interleaved.
#define VRM_DPP_STAGES(OP)                                                          \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"     \
    "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"          \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"        \
    "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"        \
    "s_nop 1"
__device__ __forceinline__ float wave_min(float v)
{
    asm volatile(VRM_DPP_STAGES("v_min_f32_dpp") : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
#undef VRM_DPP_STAGES
// lo <- min over the wave of lo, hi <- max over the wave of hi, the two chains interleaved
#define VRM_DPP_STAGE2(T, CTRL)                                       \
    "v_min_" T "_dpp %0, %0, %0 " CTRL " bank_mask:0xf\n\t"            \
    "v_max_" T "_dpp %1, %1, %1 " CTRL " bank_mask:0xf\n\t"            \
    "s_nop 0\n\t"
#define VRM_DPP_MINMAX(T)                                             \
    "s_nop 1\n\t"                                                     \
    VRM_DPP_STAGE2(T, "quad_perm:[1,0,3,2] row_mask:0xf")             \
    VRM_DPP_STAGE2(T, "quad_perm:[2,3,0,1] row_mask:0xf")             \
    VRM_DPP_STAGE2(T, "row_half_mirror row_mask:0xf")                 \
    VRM_DPP_STAGE2(T, "row_mirror row_mask:0xf")                      \
    VRM_DPP_STAGE2(T, "row_bcast:15 row_mask:0xa")                    \
    VRM_DPP_STAGE2(T, "row_bcast:31 row_mask:0xc")                    \
    "s_nop 0"
__device__ __forceinline__ void wave_minmax(float &lo, float &hi)
{
    asm volatile(VRM_DPP_MINMAX("f32") : "+v"(lo), "+v"(hi));
    lo = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, lo), 63));
    hi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, hi), 63));
}
// three (min, max) pairs at once: six independent chains, so every stage's dependent instruction is five others away
// and no wait states need padding (a chunk prologue reduces the box bounds of all three axes).  The results are left
// in lane 63 of the six registers (the caller goes on computing there and reads the integers it needs with readlane):
// the box bounds are reduced as floats and floored once, not per lane.
#define VRM_DPP_STAGE6F(CTRL)                                        \
    "v_min_f32_dpp %0, %0, %0 " CTRL " bank_mask:0xf\n\t"            \
    "v_max_f32_dpp %1, %1, %1 " CTRL " bank_mask:0xf\n\t"            \
    "v_min_f32_dpp %2, %2, %2 " CTRL " bank_mask:0xf\n\t"            \
    "v_max_f32_dpp %3, %3, %3 " CTRL " bank_mask:0xf\n\t"            \
    "v_min_f32_dpp %4, %4, %4 " CTRL " bank_mask:0xf\n\t"            \
    "v_max_f32_dpp %5, %5, %5 " CTRL " bank_mask:0xf\n\t"
__device__ __forceinline__ void wave_minmax3f_lane63(float lo[3], float hi[3])
{
    asm volatile("s_nop 1\n\t"
                 VRM_DPP_STAGE6F("quad_perm:[1,0,3,2] row_mask:0xf")
                 VRM_DPP_STAGE6F("quad_perm:[2,3,0,1] row_mask:0xf")
                 VRM_DPP_STAGE6F("row_half_mirror row_mask:0xf")
                 VRM_DPP_STAGE6F("row_mirror row_mask:0xf")
                 VRM_DPP_STAGE6F("row_bcast:15 row_mask:0xa")
                 VRM_DPP_STAGE6F("row_bcast:31 row_mask:0xc")
                 "s_nop 1"
                 : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]));
}
#undef VRM_DPP_STAGE6F
// (int)floor(x) in one instruction
__device__ __forceinline__ int floor_to_int(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}
#undef VRM_DPP_MINMAX
#undef VRM_DPP_STAGE2
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// n iterations of `dt += 0.001f; if (dt >= t_max) stop` (wgsl:181-186) with no sample in between, result-identical to
// taking them one by one.  While dt stays inside one binade and h = 0.001f is added to a multiple of the binade's ulp u,
// the sum rounds the same way every time: dt -> dt + inc with one fixed multiple inc of u (the discarded fraction of h
// is the same at every step; a fraction of exactly u/2, where round-to-even would alternate, shows as two consecutive
// increments that differ and is sent to the slow path).  dt + j*inc is then exactly representable, so j steps are ONE
// fused multiply-add, and the step that first reaches t_max is found by comparing such exact values.  Steps that cross
// into the next binade, tiny or negative dt go one at a time.  Used where a chunk has nothing to sample (empty-space
// skipping) and to bring a ray up to its rank's slab (sort-last): the iteration lattice is the oracle's, bit for bit.
__device__ __forceinline__ void advance_exact(float &dt, bool &active, float t_max, int n)
{
    {
        // the usual case, decided for the whole wave: every lane's n steps are regular (same binade to the end, see below) and
        // end short of t_max -- one FMA, nobody stops
        const float d1 = dt + VRM_STEP_SIZE, inc = d1 - dt;
        const float end = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, dt) & 0x7f800000u) + 0x00800000u);
        const float reach = __builtin_fmaf((float)n, inc, dt); // exact when the steps are regular
        const bool moves = active && n > 0;
        const bool plain = dt >= 0x1p-8f && inc > 0.0f && (d1 + VRM_STEP_SIZE) - d1 == inc && reach + VRM_STEP_SIZE < end && reach < t_max;
        if (__builtin_amdgcn_ballot_w64(moves && !plain) == 0ull) {
            dt = moves ? reach : dt;
            return;
        }
    }
    while (n > 0 && active) {
        const float d1 = dt + VRM_STEP_SIZE;
        const float inc = d1 - dt; // exact (Sterbenz)
        int k = 0;                 // steps from here that stay regular: the exact sum dt_j + h stays below the binade's end
        if (dt >= 0x1p-8f && inc > 0.0f && (d1 + VRM_STEP_SIZE) - d1 == inc) {
            const float end = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, dt) & 0x7f800000u) + 0x00800000u);
            // (j-1)*inc < end - h - dt for every regular step j; v_rcp_f32 and the roundings are covered by the 0.01 + floor
            k = (int)fminf(floorf(((end - VRM_STEP_SIZE) - dt) * __builtin_amdgcn_rcpf(inc) - 0.01f), (float)n);
        }
        if (k < 2) { // one plain step
            dt = d1;
            active = !(dt >= t_max);
            --n;
            continue;
        }
        // c = the first j >= 1 with dt + j*inc >= t_max, searched in [1, k+1]; the estimate is corrected with exact values
        float c = fminf(fmaxf(ceilf((t_max - dt) * __builtin_amdgcn_rcpf(inc)), 1.0f), (float)(k + 1));
        while (c > 1.0f && __builtin_fmaf(c - 1.0f, inc, dt) >= t_max) c -= 1.0f;
        while (c <= (float)k && __builtin_fmaf(c, inc, dt) < t_max) c += 1.0f;
        if (c <= (float)k) { // the ray ends inside this run
            dt = __builtin_fmaf(c, inc, dt);
            active = false;
            return;
        }
        dt = __builtin_fmaf((float)k, inc, dt);
        n -= k;
    }
}

// what the march kernel decides per chunk before it stages and samples (see `plan` in the kernel)
template <int R>
struct ChunkPlan {
    int n_lane[R], kk, ilo[3], ihi[3], nx, ny, nz, pitch, slice, x0u;
    bool none, fits, interior, empty;
};

// One 8x8-pixel tile of one view, marched by one wave: `xcd` and `slot` say which (see "Tile order" below).
// RAYS = 2: the tile is 16x8 pixels and a lane owns two x-adjacent pixels, whose rays it marches one after the other through
// ONE staged box per chunk.  Where a pixel is narrower than a texel (a 2048^2 view of a 1024^3 volume: half a texel) the box
// of 16x8 pixels is hardly larger than that of 8x8 -- its size is set by the slack, the +1 corner, the 16-texel staging
// quads and the obliqueness of the view -- so the chunk prologue, the occupancy look-up and the staged bytes are shared by
// twice the samples.  Which ray takes which step in which chunk does not change a ray's own dt sequence or what a sample
// contributes: the image is bit-identical to the one-ray form (tests/test_gpu_parity.py).
template <bool ALPHA_ONLY, bool USE_TF, bool DIAG, bool SKIP, int CAP, bool PACKED, bool ERT, bool FAR, int RAYS>
__device__ __forceinline__ void march_tile(const VrmMarchParams &P, const VrmViewDesc &v, const uint32_t xcd, const uint32_t slot)
{
    constexpr int R = RAYS;
    static_assert(R == 1 || R == 2, "one or two rays per lane");
    static_assert(R == 1 || !FAR, "far-camera views keep one ray per lane");
    // PACKED: the 1-channel march staged from the packed alpha bricks.  A "row unit" of the LDS tile is then a word of
    // four x-adjacent texels instead of one RGBA8 texel: a quarter of the staging bytes and of the tile's x extent.
    static_assert(!PACKED || ALPHA_ONLY, "the packed source holds the alpha channel only");
    // LDS tile layout, chosen per chunk: rows along x with pitch = the box's x extent rounded up to 4 texels
    // (16-byte staging stores), slices along z with pitch = ny * pitch rounded up to 4 (mod 8) words, which keeps
    // the 32-lane gather groups spread over the 32 banks for every view direction (DESIGN.md "LDS layout").
    constexpr int PX_MAX = 64; // one staging instruction holds at least one whole row
#ifdef VRM_AB_LDS_PAD // occupancy sensitivity experiment: same chunks, fewer resident waves
    __shared__ __attribute__((aligned(16))) uint32_t tile[CAP + VRM_AB_LDS_PAD];
#else
    __shared__ __attribute__((aligned(16))) uint32_t tile[CAP];
#endif
    const VrmGridDev &g = P.grid;
    // Tile order.  The image is cut into 64x64-pixel super-blocks (8x8 wave tiles).  Workgroup b runs on the
    // XCD group b % 8 (hardware round-robin); that group owns the super-blocks on the diagonals
    // (sbx + sby) % 8 == b % 8 and walks through them one after the other.  So the ~320 waves an XCD has in
    // flight cover a few compact image patches (=> a compact part of the volume in its L2), while every XCD
    // gets the same mix of centre (long rays) and border (misses) super-blocks: the in-order dispatcher
    // stalls on the slowest XCD, so balance matters more than anything else here.
    const uint32_t j = slot >> 6, within = slot & 63u;
    const uint32_t per_row = P.sblocks_x >> 3;             // sblocks_x is padded to a multiple of 8
    const uint32_t sby = j / per_row, m = j - sby * per_row;
    const uint32_t sbx = ((xcd - sby) & 7u) + 8u * m;
    const uint32_t tx = sbx * 8u + (within & 7u), ty = sby * 8u + (within >> 3);
    if (tx * (8u * R) >= P.width || ty * 8u >= P.height) return; // wave-uniform
    const uint32_t lane = threadIdx.x;
    const uint32_t py = ty * 8u + (lane >> 3);
    uint32_t px[R];
    bool in_image[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        px[r] = (tx * 8u + (lane & 7u)) * R + r; // (two rays: the lane's pixels are x-adjacent)
        in_image[r] = px[r] < P.width && py < P.height;
    }
    // Far cameras.  The texel coordinate u = dt*rs + bla cancels two terms of the size of bla (the camera position in
    // texels): beyond 4096 texels an ulp of them is 0.0005 texel and the one-FMA coordinate drifts away from the oracle's
    // (ro + rd*dt) * scale + bias by more than the f32 contract allows on high-contrast volumes.  Such views (hundreds of
    // volume widths away) are marched by the FAR instance of this kernel, which filters at the oracle's own operation
    // sequence -- still a function of dt alone, so every property of the chunking holds -- and gives its boxes the rounding
    // of that sequence as extra slack.  Both instances see every view of a batch; each takes the views that are its own.
    const float bla_max = fmaxf(fabsf(__builtin_fmaf(v.pos[0], g.tex_scale[0], g.tex_bias[0])),
                                fmaxf(fabsf(__builtin_fmaf(v.pos[1], g.tex_scale[1], g.tex_bias[1])),
                                      fabsf(__builtin_fmaf(v.pos[2], g.tex_scale[2], g.tex_bias[2]))));
    if ((uni(bla_max > 4096.0f ? 1 : 0) != 0) != FAR) return; // view-uniform
    Ray ray[R];
    Accum acc[R];
    uint32_t iters[R], n_inside[R], n_exec[R];
    bool active[R];
    float dt[R];
    bool any_active = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        ray[r] = setup_ray(v, g, in_image[r] ? px[r] : 0u, in_image[r] ? py : 0u, P.width, P.height);
        acc[r] = {0.0f, 0.0f, 0.0f, !ERT && VRM_TRANSMITTANCE != 0 ? 1.0f : 0.0f};
        iters[r] = n_inside[r] = n_exec[r] = 0;
        active[r] = in_image[r] && ray[r].hit;
        dt[r] = ray[r].t_min;
        any_active = any_active || active[r];
    }
    // diagnostics build: per-lane count of filtered samples whose alpha came out exactly 0, and per-wave chunk statistics
    uint32_t n_zero = 0, st_chunks = 0, st_staged = 0, st_empty = 0, st_probe_fail = 0, st_fallback = 0, st_retry = 0;
    const float kexp = -(VRM_STEP_SIZE * P.attenuation) * LOG2E * (1.0f / 255.0f) * T22; // per 2^-22 texel level
    const TfRegs tf = tf_regs<USE_TF>(P);
    if (__builtin_amdgcn_ballot_w64(any_active) == 0ull) { // the tile misses the volume (half the tiles of a sphere view): clear colour
#pragma unroll
        for (int r = 0; r < R; ++r) {
            finish_colour<ALPHA_ONLY, USE_TF, ERT>(acc[r]);
            if (in_image[r]) store_pixel<DIAG>(v, P, px[r], py, acc[r], ray[r], 0u, 0u, 0u);
        }
        return;
    }

    // Per-step advance of a ray in texel space.  The 1 % slack covers the drift of the running f32 dt sum (an ulp
    // of dt per step, 0.4 % of a step at dt = 100).
    float rs[R][3], du[R][3];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            rs[r][a] = ray[r].rd[a] * g.tex_scale[a];
            du[r][a] = rs[r][a] * (VRM_STEP_SIZE * 1.01f);
        }
    // the ray origin in absolute texel coordinates: u(dt) = dt*rs + bla is the coordinate every staged sample is filtered at
    // (the origin is the camera position: one for all rays)
    float bla[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) bla[a] = __builtin_fmaf(ray[0].ro[a], g.tex_scale[a], g.tex_bias[a]);
    constexpr bool far_cam = FAR;
    const float box_slack = 0.01f + (far_cam ? bla_max * 0x1p-20f : 0.0f);
    auto texel_coord = [&](const float dtv, float u[3], const int r) __attribute__((always_inline)) {
        if (far_cam) {
#pragma unroll
            for (int a = 0; a < 3; ++a) u[a] = __builtin_fmaf(ray[r].ro[a] + ray[r].rd[a] * dtv, g.tex_scale[a], g.tex_bias[a]);
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a) u[a] = __builtin_fmaf(dtv, rs[r][a], bla[a]);
        }
    };
    // Front alignment.  Rays start on the entry faces of the box, so "the same number of steps for every ray" keeps
    // the wave's samples on a copy of those faces: seen at a grazing angle an 8x8-pixel tile is spread over 40+
    // texels along the rays and no chunk ever fits the tile.  Instead the wave advances a plane perpendicular to its
    // major axis m: per chunk a lane takes as many steps (0..K) as bring it up to the common plane, so the box is
    // the tile's cross-section times the chunk depth whatever the entry geometry.  Which lane takes which step in
    // which chunk does not change any lane's own dt sequence: the image is bit-identical.
    // margin of the "interior chunk" test below: rounding of the one-FMA texel coordinate against the exact in-box test
    // (as integer bounds on the chunk's box, rounded outwards to whole texels)
    // Kept in ONE vector register (the kernel is short of SGPRs): lane 2a holds in_lo[a], lane 2a+1 holds -in_hi[a], so
    // that "bound k stays inside" reads (b_k ^ -(k & 1)) >= in_lim in every lane (~b = -b - 1: the upper bounds b + 1 <= in_hi).
    int in_lim;
    {
        const uint32_t ka = (lane < 6u ? lane : 0u) >> 1;
        const float ro_a = ka == 0 ? ray[0].ro[0] : ka == 1 ? ray[0].ro[1] : ray[0].ro[2];
        const float sc_a = ka == 0 ? g.tex_scale[0] : ka == 1 ? g.tex_scale[1] : g.tex_scale[2];
        const float bi_a = ka == 0 ? g.tex_bias[0] : ka == 1 ? g.tex_bias[1] : g.tex_bias[2];
        // (the ray origin is the camera position: the same for every lane)
        const float eps = 0.021f + 2e-6f * (fabsf(ro_a * sc_a) + fabsf(bi_a));
        const float lo_a = ka == 0 ? g.u_lo[0] : ka == 1 ? g.u_lo[1] : g.u_lo[2];
        const float hi_a = ka == 0 ? g.u_hi[0] : ka == 1 ? g.u_hi[1] : g.u_hi[2];
        in_lim = (lane & 1u) ? -(int)floorf(hi_a - eps) : (int)ceilf(lo_a + eps);
    }
    int m_axis = 0;
    float sgn = 1.0f, smax = 0.0f;
    bool aligned;
    {
        float amax[3], amin[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            amin[a] = __builtin_inff();
            amax[a] = -__builtin_inff();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                amin[a] = active[r] ? (R == 1 ? du[r][a] : fminf(amin[a], du[r][a])) : amin[a];
                amax[a] = active[r] ? (R == 1 ? du[r][a] : fmaxf(amax[a], du[r][a])) : amax[a];
            }
            wave_minmax(amin[a], amax[a]);
        }
        float best = -1.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float mag = fmaxf(amax[a], -amin[a]);
            if (mag > best) { best = mag; m_axis = a; sgn = amax[a] >= -amin[a] ? 1.0f : -1.0f; smax = mag; }
        }
        // every active lane must move forward along m at a comparable rate (always true for a perspective tile of a
        // sensible image; an 8x8-pixel image with a 90-degree field of view falls back to lockstep chunks)
        const float slowest = sgn > 0.0f ? (m_axis == 0 ? amin[0] : m_axis == 1 ? amin[1] : amin[2])
                                         : -(m_axis == 0 ? amax[0] : m_axis == 1 ? amax[1] : amax[2]);
        aligned = slowest > 0.25f * smax;
    }
    float rs_m[R], inv_du_m[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        rs_m[r] = sgn * (m_axis == 0 ? rs[r][0] : m_axis == 1 ? rs[r][1] : rs[r][2]);
        const float du_m = rs_m[r] * (VRM_STEP_SIZE * 1.01f); // = sgn * du[m] (a sign is exact); > 0 when aligned
        inv_du_m[r] = aligned ? 1.0f / du_m : 0.0f;
    }
    // the plane coordinate of a sample, prog(dt) = sgn * u_m(dt) = fma(dt, sgn * rs_m, sgn * bla_m) (negation commutes with
    // rounding: bit-identical), from two per-lane registers chosen once per tile -- as a select on the major axis per chunk it
    // was three FMAs, two v_cndmask and six v_readlane of spilled lane masks
    const float bla_m = sgn * (m_axis == 0 ? bla[0] : m_axis == 1 ? bla[1] : bla[2]);
    // Boxes from the tile's geometry.  The rays of a tile are a pencil through the camera: at plane coordinate p (along the
    // major axis, see `prog` below) lane l is at u_a = bla_a + (p - p_cam) * B_l[a], B_l[a] = rs_l[a] / |rs_l[m]|, and a
    // chunk's samples lie between the plane of the rear-most lane and the plane the chunk ends at.  With the extremes of B
    // over the tile's rays (reduced ONCE per tile) the box of a chunk is six fused multiply-adds on wave-uniform numbers --
    // lane 2a holds the lower bound of axis a, lane 2a+1 the upper -- instead of a six-chain reduction over the lanes' own
    // first and last positions per chunk.  A bound, not the hull: lanes that have ended still count, and the rounding of
    // this formula against the samples' own u(dt) (about 1e-3 texel at 4096 texels from the camera) is covered by the slack.
#ifndef VRM_GEO_BOX
#define VRM_GEO_BOX 1
#endif
    constexpr bool GEO = VRM_GEO_BOX != 0 && !FAR;
    // (All of it lives in VGPRs -- the kernel is short of SGPRs: geo_bn / geo_bf are the slope B where the near / the far
    // plane gives this lane's bound and 0 where the other one does, geo_c the bound at p = 0 with the slack, p_cam the
    // camera's plane coordinate, +inf where this path is off.)
    float geo_bn = 0.0f, geo_bf = 0.0f, geo_c = 0.0f, geo_cl = 0.0f, geo_ch = 0.0f, p_cam = __builtin_inff();
    if (GEO && aligned) {
        float inv_m[R];
#pragma unroll
        for (int r = 0; r < R; ++r) inv_m[r] = sgn / (m_axis == 0 ? rs[r][0] : m_axis == 1 ? rs[r][1] : rs[r][2]); // 1 / |rs_m|: every active lane moves forward
        float bmn[3], bmx[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            bmn[a] = __builtin_inff();
            bmx[a] = -__builtin_inff();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float b = rs[r][a] * inv_m[r];
                bmn[a] = active[r] ? (R == 1 ? b : fminf(bmn[a], b)) : bmn[a];
                bmx[a] = active[r] ? (R == 1 ? b : fmaxf(bmx[a], b)) : bmx[a];
            }
            wave_minmax(bmn[a], bmx[a]);
        }
        const uint32_t k = lane < 6u ? lane : 0u, ka = k >> 1;
        const bool upper = (k & 1u) != 0u;
        const float geo_b = ka == 0 ? (upper ? bmx[0] : bmn[0]) : ka == 1 ? (upper ? bmx[1] : bmn[1]) : (upper ? bmx[2] : bmn[2]);
        p_cam = sgn * (m_axis == 0 ? bla[0] : m_axis == 1 ? bla[1] : bla[2]);
        // u = bla + (p - p_cam) * B = (bla - p_cam * B) + p * B: the first term rounds by an ulp of the camera's distance in
        // texels (2.4e-4 at 4096), inside the slack
        geo_c = __builtin_fmaf(-p_cam, geo_b, ka == 0 ? bla[0] : ka == 1 ? bla[1] : bla[2]) + (upper ? box_slack : -box_slack);
        const bool geo_far = (geo_b >= 0.0f) == upper; // which of the two planes gives this bound
        geo_bn = geo_far ? 0.0f : geo_b;
        geo_bf = geo_far ? geo_b : 0.0f;
        // (the clamps of chunk_box below: a box outside the volume (slab) degenerates)
        geo_cl = upper ? -0x1p28f : (ka == 0 ? g.u_lo[0] : ka == 1 ? g.u_lo[1] : g.u_lo[2]) - 0.5f;
        geo_ch = upper ? (ka == 0 ? g.u_hi[0] : ka == 1 ? g.u_hi[1] : g.u_hi[2]) : 0x1p28f;
    }
    int steps_left[R]; // wgsl:139: at most 5000 iterations per ray
#pragma unroll
    for (int r = 0; r < R; ++r) steps_left[r] = VRM_MAX_STEP_AMOUNT;
    // Empty space in long strides.  After an empty chunk the next one is tried 2^lvl times as deep and tested against level
    // lvl of the occupancy pyramid (cells 2^lvl times as wide: as many look-ups as a plain chunk); a deeper chunk that is not
    // empty costs one prologue and drops back to lvl 0, where chunks are staged and sampled.  How the march is cut into
    // chunks does not change what a sample contributes (see the steps below), so this is as exact as plain skipping.
    int lvl = SKIP && VRM_SKIP_START_LEVEL > 0 && g.occ_off[VRM_SKIP_START_LEVEL] != 0 ? VRM_SKIP_START_LEVEL : 0, lvl_run = 0;
    // byte offset of pyramid level L in lane L of one register (indexed by the level as an array of the kernel argument it is a
    // scalar load and its round trip in every chunk prologue)
    const uint32_t occ_off_v = lane == 1u ? g.occ_off[1] : lane == 2u ? g.occ_off[2] : lane == 3u ? g.occ_off[3] : 0u;
    const uint32_t lane_c0 = lane & 3u, lane_c1 = (lane >> 2) & 3u, lane_c2 = lane >> 4; // this lane's cell in an emptiness test

    // Per-wave chunk depth.  How deep a box fits the tile depends on the view direction through this tile (an oblique
    // tile's box is wide on all three axes): a box that had to be cut down sets the depth for the chunks that follow
    // (no second box computation per chunk), and a box that leaves a fifth of the tile unused lets it grow by an eighth,
    // up to 1.5x the launch's depth.  Chunking does not change the image (see the steps below).
    int kk_base = (int)P.chunk_steps;
    // One chunk's plan: steps per lane, the box, how it lies in the tile, whether it is empty.  A function of the ray state
    // (dt, active, steps left) and the stride level alone, so it can be made for the NEXT chunk while the current chunk's
    // box is still in flight (look-ahead below).
    auto plan = [&](const float (&dt)[R], const bool (&active)[R], const int (&steps_left)[R], const int lvl, ChunkPlan<R> &q) __attribute__((always_inline)) {
        float ua[R][3];
        float prog[R];
        float prog_min = __builtin_inff();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            ua[r][0] = ua[r][1] = ua[r][2] = 0.0f;
            if (far_cam) {
                texel_coord(dt[r], ua[r], r); // (the formula the samples are filtered at)
                prog[r] = sgn * (m_axis == 0 ? ua[r][0] : m_axis == 1 ? ua[r][1] : ua[r][2]);
            } else {
                prog[r] = __builtin_fmaf(dt[r], rs_m[r], bla_m);
            }
            prog_min = active[r] ? (R == 1 ? prog[r] : fminf(prog_min, prog[r])) : prog_min;
        }
        const float rear = aligned ? wave_min(prog_min) : 0.0f;
        // chunk length = the steps of the rear-most lane: the wave's learned depth for plain chunks, strides in empty space
        int kk = lvl > 0 ? (int)P.chunk_steps << lvl : kk_base;
        int (&n_lane)[R] = q.n_lane; // steps of this lane's rays in this chunk
        int *ilo = q.ilo, *ihi = q.ihi;
        int &nx = q.nx, &ny = q.ny, &nz = q.nz, &pitch = q.pitch, &slice = q.slice;
        int &x0u = q.x0u; // first staged column in row units (texels; words of 4 texels when PACKED)
        bool &none = q.none, &fits = q.fits, &interior = q.interior;
        // texel-space bounding box [lo, hi] of everything `nl` steps of every lane can touch; returns this lane's
        // "every step stays inside the volume (slab) with margin"
        auto chunk_box = [&](const int (&nl)[R], int lo[3], int hi[3]) __attribute__((always_inline)) -> bool {
            // texels floor(u) and floor(u)+1 are read; samples outside [-0.5, dim-0.5) fail the in-box test.  Per lane only
            // the float extent of its steps; the wave's extent is reduced as floats and floored once, in lane 63, so the
            // bounds land in SGPRs as integers.  (z of a slab-partitioned volume: samples outside the slab's own range fail
            // the test too, so a chunk that lies in another rank's slab degenerates to `none` and only advances dt.)
            float mn[3], mx[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                mn[a] = __builtin_inff();
                mx[a] = -__builtin_inff();
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool has = nl[r] > 0;
                const float nlf = (float)nl[r];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float ue = __builtin_fmaf(nlf, du[r][a], ua[r][a]);
                    if (R == 1) {
                        mn[a] = has ? fminf(ua[r][a], ue) : __builtin_inff();
                        mx[a] = has ? fmaxf(ua[r][a], ue) : -__builtin_inff();
                    } else {
                        mn[a] = has ? fminf(mn[a], fminf(ua[r][a], ue)) : mn[a];
                        mx[a] = has ? fmaxf(mx[a], fmaxf(ua[r][a], ue)) : mx[a];
                    }
                }
            }
            wave_minmax3f_lane63(mn, mx);
            bool inter = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                // (a wave without steps: +-inf clamp to +-2^28, an empty box)
                const int li = __builtin_amdgcn_readlane(floor_to_int(__builtin_amdgcn_fmed3f(mn[a] - box_slack, g.u_lo[a] - 0.5f, 0x1p28f)), 63);
                const int hh = __builtin_amdgcn_readlane(floor_to_int(__builtin_amdgcn_fmed3f(mx[a] + box_slack, -0x1p28f, g.u_hi[a])), 63) + 1;
                // interior: min - 0.01 >= li >= in_lo >= u_lo + eps and max + 0.01 < hh <= in_hi <= u_hi - eps
                inter = inter && li >= __builtin_amdgcn_readlane(in_lim, 2 * a) && hh <= -__builtin_amdgcn_readlane(in_lim, 2 * a + 1);
                lo[a] = li;
                hi[a] = hh;
            }
            return inter;
        };
        // ---- exact empty-space skipping: if every texel the chunk can touch has alpha == 0, every sample has
        // a == 0, alpha_src = 1 - exp(0) = 0 exactly and neither colour nor alpha changes (wgsl:150-159); only
        // the dt lattice has to advance.  A look-up in level lvl of the occupancy pyramid over the cells the box [ilo, ihi]
        // overlaps (up to two per lane); a box of more cells counts as not empty.
        auto box_is_empty = [&]() __attribute__((always_inline)) -> bool {
            const int sh = 2 + lvl; // cells of level lvl are 4 << lvl texels wide
            const int mxx = (int)g.dim[0] - 1, mxy = (int)g.dim[1] - 1, mxz = (int)g.dim[2] - 1;
            const int c0x = max(ilo[0], 0) >> sh, c0y = max(ilo[1], 0) >> sh, c0z = max(ilo[2], 0) >> sh;
            const int ncx = (min(ihi[0], mxx) >> sh) - c0x + 1, ncy = (min(ihi[1], mxy) >> sh) - c0y + 1,
                      ncz = (min(ihi[2], mxz) >> sh) - c0z + 1;
            const uint32_t rnd = (1u << lvl) - 1u;
            const uint32_t od0 = (g.occ_dim[0] + rnd) >> lvl, od1 = (g.occ_dim[1] + rnd) >> lvl;
            const uint8_t *occl = g.occ + (uint32_t)__builtin_amdgcn_readlane((int)occ_off_v, uni(lvl));
            if (ncx <= 4 && ncy <= 4 && ncz <= 4) {
                // the usual case: the cells are a corner of the 4x4x4 block the lanes stand for (lane = x + 4*y + 16*z), one
                // byte per lane at a lane-constant offset from the block's first cell (a level fits 32-bit offsets)
                const uint32_t first = ((uint32_t)c0z * od1 + (uint32_t)c0y) * od0 + (uint32_t)c0x;
                const uint32_t off = first + (lane_c2 * od1 + lane_c1) * od0 + lane_c0;
                const bool mine = (int)lane_c0 < ncx && (int)lane_c1 < ncy && (int)lane_c2 < ncz;
                return __builtin_amdgcn_ballot_w64(mine && occl[off] != 0) == 0ull;
            }
            const int ncxy = ncx * ncy, nc = ncxy * ncz;
            if (nc > 128) return false;
            const float inv_xy = __builtin_amdgcn_rcpf((float)ncxy), inv_x = __builtin_amdgcn_rcpf((float)ncx);
            bool occ = false;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int c = (int)lane + 64 * r;
                if (c < nc) {
                    const int cz = (int)(((float)c + 0.5f) * inv_xy), rem = c - cz * ncxy;
                    const int cy = (int)(((float)rem + 0.5f) * inv_x), cx = rem - cy * ncx;
                    occ |= occl[((size_t)(c0z + cz) * od1 + (size_t)(c0y + cy)) * (size_t)od0 + (size_t)(c0x + cx)] != 0;
                }
            }
            return __builtin_amdgcn_ballot_w64(occ) == 0ull;
        };
        // the same box from the tile's geometry (see geo_b above): every sample of a chunk of depth k lies between the planes
        auto geo_box = [&](int k, int lo[3], int hi[3]) __attribute__((always_inline)) -> bool {
            const float far = __builtin_fmaf((float)k + 0.5f, smax, rear);
            const int iv = floor_to_int(__builtin_amdgcn_fmed3f(__builtin_fmaf(rear, geo_bn, __builtin_fmaf(far, geo_bf, geo_c)), geo_cl, geo_ch));
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = __builtin_amdgcn_readlane(iv, 2 * a);
                hi[a] = __builtin_amdgcn_readlane(iv, 2 * a + 1) + 1;
            }
            // interior, all six bounds at once (see in_lim)
            const int flip = -(int)(lane & 1u);
            return (__builtin_amdgcn_ballot_w64((iv ^ flip) >= in_lim) & 0x3full) == 0x3full;
        };
        // the plane sits (kk + 1/2) of the fastest lane's steps ahead of the rear-most lane: that lane takes kk steps
        auto lane_steps = [&](int k, const int r) __attribute__((always_inline)) -> int {
            const float ahead = aligned ? (rear + ((float)k + 0.5f) * smax - prog[r]) * inv_du_m[r] : (float)k;
            return active[r] ? min((int)__builtin_amdgcn_fmed3f(ahead, 0.0f, (float)k), steps_left[r]) : 0;
        };
        // (a camera inside the volume marches from t_min < 0, behind itself, where (p - p_cam) * B changes its sense)
        // (Not for a slab-partitioned volume either: its waves are active in part for much of their way -- rays enter and
        // leave the slab at different planes -- and the hull of the lanes that still step is the tighter box there: 2.27
        // against 2.39 ms/view for the C5 rank of profiles/r2.  Tested here, where g.slab is at hand anyway: a test at the
        // tile's set-up costs the C3 instance ten more spilled SGPRs and 3 % of its time.)
        const bool use_geo = GEO && !g.slab && __builtin_amdgcn_ballot_w64(!(rear >= p_cam)) == 0ull;
        bool empty = false;
        for (;;) {
            if (use_geo) interior = geo_box(kk, ilo, ihi);
            else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (!far_cam) texel_coord(dt[r], ua[r], r); // (only the reduction over the lanes' own positions needs all three axes)
                    n_lane[r] = lane_steps(kk, r);
                }
                interior = chunk_box(n_lane, ilo, ihi); // every step of every lane stays inside the volume (slab) with margin
            }
            none = ihi[0] - ilo[0] < 1 || ihi[1] - ilo[1] < 1 || ihi[2] - ilo[2] < 1; // whole chunk lies outside the box
            // rows start on a 4-unit boundary: the staging loads are whole 16-byte quads of the bricked volume
            x0u = PACKED ? ilo[0] >> 2 : ilo[0];
            if (x0u >= 0) x0u &= ~3;
            if (!PACKED) ilo[0] = x0u;
            nx = (PACKED ? ihi[0] >> 2 : ihi[0]) - x0u + 1; ny = ihi[1] - ilo[1] + 1; nz = ihi[2] - ilo[2] + 1;
            pitch = (nx + 3) & ~3;
            slice = ny * pitch;
            slice += (slice & 4) ? 0 : 4;
            // (PACKED reads the word after a sample's word as well: keep one word of the tile spare)
            // (ny, nz <= CAP first: the clamped bounds span up to 2^29 texels, and pitch <= 64, so the product then fits 32 bits)
            fits = !none && pitch <= PX_MAX && ny <= CAP && nz <= CAP && nz * slice + (PACKED ? 1 : 0) <= CAP;
            if (fits || none || kk == 1 || lvl > 0) break; // (a deep empty-space probe is never staged: it need not fit)
            kk = (3 * kk) >> 2; // 3/4 steps down to 1 (2 -> 1 as 6 >> 2)
            if (DIAG) ++st_retry;
        }
        if (use_geo) {
#pragma unroll
            for (int r = 0; r < R; ++r) n_lane[r] = lane_steps(kk, r);
        }
        // Progress guarantee.  The rear-most active lane always takes kk >= 1 steps -- unless its ray parameter is not a number:
        // aabb_intersect divides by the ray direction (wgsl:247-299), and a camera far enough away that f32 leaves its rays
        // parallel to an axis, or one that stands in a face plane of the box, gets t_min = NaN or -inf for some pixels.  The
        // reference marches such a ray through its 5000 iterations without a sample (every comparison of the in-box test is
        // false); here every step count derived from its position is 0, and once only such lanes are left the wave would plan
        // the same chunk for ever (found with a camera 1000 box widths away, tools/far_one.py).  A wave none of whose lanes
        // would step takes lockstep chunks of 8 steps through the global-gather path, whose per-sample in-box test is the
        // oracle's own: it ends with the iteration count and the (empty) result of the reference.
        bool stuck = true;
#pragma unroll
        for (int r = 0; r < R; ++r) stuck = stuck && n_lane[r] <= 0;
        stuck = __builtin_amdgcn_ballot_w64(!stuck) == 0ull;
        if (stuck) { // (no box, no emptiness test: the gather path needs neither)
            kk = 8;
#pragma unroll
            for (int r = 0; r < R; ++r) n_lane[r] = active[r] ? min(8, steps_left[r]) : 0;
            none = false;
            fits = false;
            interior = false;
        } else if (none) empty = true;
        else {
            if (!fits && lvl == 0) { // global-gather fallback chunk: lockstep, 8 steps
                kk = 8;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    n_lane[r] = active[r] ? min(8, steps_left[r]) : 0;
                    // the box above is that of the ONE step that did not fit; the emptiness test must cover all 8
                    if (!far_cam) texel_coord(dt[r], ua[r], r);
                }
                (void)chunk_box(n_lane, ilo, ihi);
            }
            if (SKIP) empty = box_is_empty();
        }
        q.kk = kk;
        q.empty = empty;
    };
    ChunkPlan<R> cp;
#ifdef VRM_LOOKAHEAD
    static_assert(R == 1, "the look-ahead experiment is a one-ray build");
    ChunkPlan<R> cp_next;
    bool have_next = false;
#endif
    for (;;) {
        {
            bool any = false;
#pragma unroll
            for (int r = 0; r < R; ++r) any = any || active[r];
            if (__builtin_amdgcn_ballot_w64(any) == 0ull) break;
        }
        // Issue priority by phase: a wave in its step loop goes before a wave that plans or stages (s_setprio 1 / 0).  The step
        // loops are what the SIMD's VALU is short of; a planning wave mostly waits on reductions and memory anyway.
        // C3 0.367 -> 0.359 ms/view, C4 1.088 -> 1.057; the level (1..3) makes no difference, the other polarity loses, and so does
        // raising the priority already for the staging loads (+1..2 %).
        __builtin_amdgcn_s_setprio(0);
#ifdef VRM_LOOKAHEAD
        if (have_next) {
            cp = cp_next;
            have_next = false;
        } else
#endif
        {
            // (two rays per lane: the launcher keeps slab-partitioned volumes on the one-ray form)
            if (R == 1 && !DIAG && g.slab) {
                // Slab-partitioned volume: this rank marches the whole volume's lattice but blends only its slab's samples.
                // (The diagnostics build walks every step so that its per-ray counts stay those of the unpartitioned march.)
                const float uz = __builtin_fmaf(dt[0], rs[0][2], bla[2]), dz = du[0][2], lo1 = g.u_lo[2] - 1.0f, hi1 = g.u_hi[2] + 1.0f; // one texel of margin
                // past the slab and moving away: nothing left to blend, the accumulators are final
                if (dz > 0.0f ? uz >= hi1 : (dz < 0.0f && uz < lo1)) active[0] = false;
                if (__builtin_amdgcn_ballot_w64(active[0]) == 0ull) break;
                // before the slab: advance the running dt sum up to the slab's entry plane without a chunk prologue per K
                // steps (du carries 1 % slack, so the step count errs on the short side)
                float ff = 0.0f;
                if (dz > 0.0f && uz < lo1) ff = (lo1 - uz) / dz;
                else if (dz < 0.0f && uz >= hi1) ff = (hi1 - uz) / dz;
                const int nff = uni((int)fminf(wave_min(active[0] ? ff : __builtin_inff()), (float)VRM_MAX_STEP_AMOUNT)) - 1;
                if (nff > 0) {
                    advance_exact(dt[0], active[0], ray[0].t_max, nff);
                    steps_left[0] -= nff;
                    active[0] = active[0] && steps_left[0] > 0;
                    continue;
                }
            }
            plan(dt, active, steps_left, lvl, cp);
        }
        int (&n_lane)[R] = cp.n_lane;
        const int *ilo = cp.ilo;
        const int nx = cp.nx, ny = cp.ny, nz = cp.nz, pitch = cp.pitch, slice = cp.slice, x0u = cp.x0u, kk = cp.kk;
        const bool fits = cp.fits, interior = cp.interior;
        bool empty = cp.empty;
        if (lvl == 0 && fits) {
            if (kk < kk_base) kk_base = kk;
            else if (nz * slice * 5 < CAP * 4 && 2 * kk_base < 3 * (int)P.chunk_steps) kk_base += max(1, kk_base >> 3);
        }
        if (DIAG) ++st_chunks;
        if (SKIP && lvl > 0 && !empty) { // the deep probe found something (or could not be tested): back to plain chunks
            lvl = VRM_SKIP_FALLBACK ? lvl - 1 : 0; // (A/B: one level down instead -- the level that was last found empty)
            lvl_run = VRM_SKIP_FALLBACK ? -VRM_SKIP_FALLBACK : 0; // (... and that many more empty chunks before the next probe)
            if (DIAG) ++st_probe_fail;
            continue; // nothing was advanced
        }

#ifdef VRM_TIMING_NO_STAGE // timing build: prologue + dt advance only
        empty = true;
#endif
        if (fits && !empty) {
            // ---- stage the box: rows along x, clamp-to-edge applied here --------------------------
            const int rows = ny * nz;
            const float inv_ny = __builtin_amdgcn_rcpf((float)ny);
            const int mxx = (int)g.dim[0] - 1, mxy = (int)g.dim[1] - 1, mxz = (int)g.dim[2] - 1;
            uint32_t seen = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // previous chunk's reads before these writes
            // lane / n for a wave-uniform n <= 64: (lane + 0.5)/n is at least 0.5/n away from an integer, far above the error of
            // v_rcp_f32 (1 ulp; a correctly rounded division costs 12 VALU instructions here, once per chunk)
            const uint32_t *src = PACKED ? g.abricks : g.bricks; // both edge-replicated up to their brick boundaries
            const uint32_t src_nbx = PACKED ? g.nbxa : g.nbx;
            if (x0u >= 0 && x0u + pitch <= (PACKED ? (int)(g.nbxa * 4u) : (int)g.dim[0])) {
                // rows lie inside the volume in x.  Source: the bricked copy; transport: LDS-DMA (global_load_lds_dwordx4),
                // 16 bytes per lane straight into the tile, no VGPR round trip, no ds_write, every load of the box in
                // flight at once.  One instruction fills up to 64 consecutive quads of one z slice: lane -> (x quad q,
                // row ey) with LDS word ey*pitch + 4q = 4*lane, exactly the lane-linear image the DMA writes.  The per-lane
                // source offset (brick column and row) is the same for every slice; the slice offset is scalar.
                const int qpr = pitch >> 2;
                const int rpw = (int)(64.5f * __builtin_amdgcn_rcpf((float)qpr)); // = 64 / qpr for qpr in 1..16 (the half absorbs the 1-ulp rcp)
                const int eyl = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)qpr)), q = (int)lane - eyl * qpr;
                const uint32_t bx = ((uint32_t)x0u >> 2) + (uint32_t)q;
                const size_t zslab = (size_t)src_nbx * g.nby * 128u;   // bytes per brick layer (2 slices)
                // The usual box lies inside the volume in z as well: no clamp per slice, and the slices of one brick layer are
                // 64 bytes apart.  Written out (one s_mov m0 + the load per slice, one 64-bit scalar add per layer, the lane mask set
                // once per row block): the compiler's form of the general loop below spends 20 scalar instructions per load on
                // the clamp and a 64-bit multiply-add, and the CU's one scalar unit issues 0.95 instructions per cycle for all
                // its waves (tools/probes/salu_probe.hip) -- it was as busy as the vector units (C4: 0.70 per CU-cycle).
                const bool z_plain = ilo[2] >= 0 && ilo[2] + nz - 1 <= mxz && zslab < (1ull << 31);
                for (int eb = 0; eb * rpw < ny; ++eb) {
                    const int ey = eb * rpw + eyl;
                    const bool lane_on = eyl < rpw && ey < ny;
                    const uint32_t gy = (uint32_t)min(max(ilo[1] + ey, 0), mxy);
                    const uint32_t voff = (((gy >> 2) * src_nbx + bx) * 8u + (gy & 3u)) * 16u;
                    uint32_t *drow = &tile[eb * rpw * pitch];
                    if (z_plain) {
                        if (lane_on) {
                            auto dma = [](uint32_t lds_byte, uint32_t off, const char *base) __attribute__((always_inline)) {
                                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_byte), "v"(off), "s"(base) : "memory");
                            };
                            uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t *)drow;
                            const uint32_t lds_step = (uint32_t)slice * 4u, voff1 = voff + 64u;
                            const char *lay = reinterpret_cast<const char *>(src) + (size_t)((uint32_t)ilo[2] >> 1) * zslab;
                            int ez = 0;
                            if (ilo[2] & 1) {
                                dma(lds, voff1, lay);
                                lay += zslab;
                                lds += lds_step;
                                ez = 1;
                            }
                            for (; ez + 1 < nz; ez += 2) {
                                dma(lds, voff, lay);
                                dma(lds + lds_step, voff1, lay);
                                lay += zslab;
                                lds += 2u * lds_step;
                            }
                            if (ez < nz) dma(lds, voff, lay);
                        }
                        continue;
                    }
                    for (int ez = 0; ez < nz; ++ez) {
                        const uint32_t gz = (uint32_t)min(max(ilo[2] + ez, 0), mxz); // uniform
                        const char *zbase = reinterpret_cast<const char *>(src) + (size_t)(gz >> 1) * zslab + (gz & 1u) * 64u;
                        if (lane_on)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(zbase + voff),
                                                             (__attribute__((address_space(3))) void *)(drow + ez * slice), 16, 0, 0);
                    }
                }
#ifdef VRM_LOOKAHEAD
                // Look-ahead: while the box is in flight, plan the next chunk.  Without the early exit the state after this
                // chunk's steps is known before they run (the dt lattice and the geometric end of the ray: advance_exact), and
                // a staged chunk is followed by a plain one (lvl 0).  The plan's occupancy look-up queues behind the DMA.
                if (!ERT && !DIAG && !g.slab) {
                    float dt1[1] = {dt[0]};
                    bool a1[1] = {active[0]};
                    advance_exact(dt1[0], a1[0], ray[0].t_max, n_lane[0]);
                    const int sl1[1] = {steps_left[0] - n_lane[0]};
                    a1[0] = a1[0] && sl1[0] > 0;
                    if (__builtin_amdgcn_ballot_w64(a1[0]) != 0ull) {
                        plan(dt1, a1, sl1, 0, cp_next);
                        have_next = true;
                    }
                }
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                seen = 0xff000000u; // the staged texels do not pass through registers: emptiness comes from the occupancy grid alone
            } else {
                // the box sticks out of the volume in x: one texel per lane, clamp-to-edge per texel
                const int rpi = (int)(64.5f * __builtin_amdgcn_rcpf((float)pitch));
                const int er = (int)(((float)lane + 0.5f) * __builtin_amdgcn_rcpf((float)pitch)), ex = (int)lane - er * pitch;
                const int tx0 = PACKED ? 4 * (x0u + ex) : x0u + ex; // first texel of this lane's row unit
                const uint32_t gx = (uint32_t)min(max(tx0, 0), mxx);
                const bool lane_on = er < rpi && ex < nx;
#pragma unroll 4
                for (int row = er; row < rows; row += rpi) {
                    // row -> (ey, ez)
                    const int ez = (int)(((float)row + 0.5f) * inv_ny);
                    const int ey = row - ez * ny;
                    const uint32_t gy = (uint32_t)min(max(ilo[1] + ey, 0), mxy);
                    const uint32_t gz = (uint32_t)min(max(ilo[2] + ez, 0), mxz);
                    if (lane_on) {
                        uint32_t w;
                        if (PACKED) {
                            // alpha of texels tx0 .. tx0+3, each clamped to the edge: a word that starts inside the padded brick
                            // range is the stored word (texels past the volume repeat the edge there), a word outside it is
                            // four copies of the edge texel -- from the packed bricks alone (a scalar volume keeps nothing else)
                            if (tx0 >= 0 && tx0 < (int)(g.nbxa * 16u)) w = g.abricks[abrick_word(g, (uint32_t)tx0 >> 2, gy, gz)];
                            else w = abrick_alpha(g, gx, gy, gz) * 0x01010101u;
                            seen |= w ? 0xff000000u : 0u;
                        } else {
                            w = g.vox[((size_t)gz * g.dim[1] + gy) * g.dim[0] + gx];
                            seen |= w;
                        }
                        tile[ez * slice + ey * pitch + ex] = w;
                    }
                }
            }
            if (SKIP) empty = __builtin_amdgcn_ballot_w64((seen >> 24) != 0u) == 0ull; // staged box is the exact touch set
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // writes before the gathers below
            __builtin_amdgcn_wave_barrier();
        }

#ifdef VRM_TIMING_NO_STEPS // timing build: prologue + staging + dt advance, no samples
        empty = true;
#endif
        if (DIAG) {
            if (empty) ++st_empty;
            else if (fits) ++st_staged;
            else ++st_fallback;
        }
        if (empty) {
            // nothing to sample in this chunk: advance the running dt sum exactly as the loop would (wgsl:181-186)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (!DIAG) advance_exact(dt[r], active[r], ray[r].t_max, n_lane[r]); // (the diagnostics build counts in-box samples per step)
                else
                for (int j = 0; j < kk; ++j) {
                    if (j < n_lane[r] && active[r]) {
                        if (DIAG) {
                            ++iters[r];
                            const float p0 = ray[r].ro[0] + ray[r].rd[0] * dt[r], p1 = ray[r].ro[1] + ray[r].rd[1] * dt[r], p2 = ray[r].ro[2] + ray[r].rd[2] * dt[r];
                            n_inside[r] += ((p0 >= g.box_min[0]) & (p1 >= g.box_min[1]) & (p2 >= g.in_min_z) & (p0 < g.box_max[0]) &
                                            (p1 < g.box_max[1]) & (p2 < g.in_max_z)) ? 1u : 0u;
                        }
                        dt[r] += VRM_STEP_SIZE;
                        active[r] = !(dt[r] >= ray[r].t_max);
                    }
                }
                steps_left[r] -= n_lane[r]; // a lane that stopped early is inactive anyway
                active[r] = active[r] && steps_left[r] > 0;
            }
            if (SKIP && VRM_SKIP_MAX_LEVEL > 0) {
                // (level L exists only if the pyramid fits 32-bit offsets: occ_off[L] != 0)
                if (++lvl_run >= VRM_SKIP_ESCALATE && lvl < VRM_SKIP_MAX_LEVEL && __builtin_amdgcn_readlane((int)occ_off_v, uni(lvl + 1)) != 0) { ++lvl; lvl_run = 0; }
            }
            continue;
        }
        lvl_run = 0;
        __builtin_amdgcn_s_setprio(1);

        // ---- the chunk's steps: lane l takes n_lane of them -------------------------------------------
        if (fits) {
            // Texel coordinate of a sample.  The sample POSITION p = ro + rd*dt and the in-box test must be the
            // oracle's exact operation sequence, but they only matter where a sample can fall outside the box: a chunk
            // whose whole box lies inside the volume with margin ("interior", the common case) skips both.  The
            // coordinate that feeds the filter -- continuous arithmetic -- is u = dt*(rd*scale) + (ro*scale + bias), one
            // FMA per axis, in ABSOLUTE texel units and by the same formula in interior and border chunks; floor(u) and
            // the weights u - floor(u) come from it, and only the integer cell index is made relative to the staged box
            // (exact).  So what a sample contributes does not depend on how the march was cut into chunks or where a
            // chunk's box starts: any chunking, data-dependent ones included, gives bit-identical accumulators.
            const float o1 = (float)ilo[1], o2 = (float)ilo[2];
            const float fpx4 = (float)(4 * pitch), fsl4 = (float)(4 * slice);
            const float x04 = -4.0f * (float)x0u; // first staged column in row units (texels; words of 4 texels when PACKED), in bytes
            const int pitch4 = 4 * pitch, slice4 = 4 * slice;
            // byte address of the last corner cell (PACKED: rows of 4 * nx texel bytes, the cell's x + 1 is the last one)
            const float amax4 = (float)(4 * ((nz - 2) * slice + (ny - 2) * pitch) + (PACKED ? 4 * nx - 2 : 4 * (nx - 2)));
            const char *tile_b = reinterpret_cast<const char *>(tile);
            auto ld = [&](int byte_off) -> uint32_t { return *reinterpret_cast<const uint32_t *>(tile_b + byte_off); };
            auto steps = [&](auto interior_c, auto ray_c) {
                constexpr bool INTERIOR = decltype(interior_c)::value;
                constexpr int r = decltype(ray_c)::value;
                // (The lane's state inside the loop is ONE vector register: the step it stops at, -1 once the ray has ended.  As
                // `j < n_lane && active` with the mask `active` updated per step the loop carried 20 scalar instructions per
                // sample -- see the staging loop above for what they cost.)
                int lim = active[r] ? n_lane[r] : 0;
                const int kku = uni(kk); // (wave-uniform by construction; said so, the loop control is scalar)
                for (int j = 0; j < kku; ++j) {
                    if (j < lim) {
                        if (DIAG) ++iters[r];
                        bool inside = true;
                        if constexpr (!INTERIOR) {
                            const float p0 = ray[r].ro[0] + ray[r].rd[0] * dt[r];
                            const float p1 = ray[r].ro[1] + ray[r].rd[1] * dt[r];
                            const float p2 = ray[r].ro[2] + ray[r].rd[2] * dt[r];
                            inside = (p0 >= g.box_min[0]) & (p1 >= g.box_min[1]) & (p2 >= g.in_min_z) & (p0 < g.box_max[0]) &
                                     (p1 < g.box_max[1]) & (p2 < g.in_max_z);
                        }
                        bool done = false;
                        if (inside) {
                            if (DIAG) { ++n_inside[r]; ++n_exec[r]; }
                            float uu[3];
                            if constexpr (FAR) {
#pragma unroll
                                for (int a = 0; a < 3; ++a) uu[a] = __builtin_fmaf(ray[r].ro[a] + ray[r].rd[a] * dt[r], g.tex_scale[a], g.tex_bias[a]);
                            } else {
#pragma unroll
                                for (int a = 0; a < 3; ++a) uu[a] = __builtin_fmaf(dt[r], rs[r][a], bla[a]);
                            }
                            const float u0 = uu[0], u1 = uu[1], u2 = uu[2];
                            const float f0 = floorf(u0), f1 = floorf(u1), f2 = floorf(u2);
                            // the box holds floor(u) and floor(u)+1 of every in-box sample by construction (0.01 texel
                            // of slack); the clamp is insurance against a rounding surprise, the weights stay unclamped
                            // (PACKED: the tile's rows are byte-linear in x -- byte x - 4 * x0u of the row is texel x)
                            const float af = __builtin_fmaf(f2 - o2, fsl4, __builtin_fmaf(f1 - o1, fpx4, PACKED ? f0 + x04 : __builtin_fmaf(f0, 4.0f, x04))); // exact: small integers
                            // (interior chunks: every sample lies inside the volume and the box holds its cell by construction --
                            // the geometric bound with its 0.01-texel slack -- so the clamp, one half-rate instruction per sample,
                            // is left to the border chunks, where a sample rejected by the in-box test has no cell in the box)
                            const int a0 = INTERIOR && VRM_INTERIOR_NO_CLAMP ? (int)af : (int)__builtin_amdgcn_fmed3f(af, 0.0f, amax4);
                            const int a1 = a0 + pitch4, a2 = a0 + slice4, a3 = a2 + pitch4;
                            if constexpr (PACKED) {
                                // (written out: the compiler merges the byte pair of a row into one ds_read_u16, and a halfword read
                                // at an odd address, which every other sample's is, takes six times as long -- valu_probe2)
                                uint32_t t[8];
                                const uint32_t tb = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char *)tile_b;
                                asm volatile("ds_read_u8 %0, %8\n\tds_read_u8 %1, %8 offset:1\n\tds_read_u8 %2, %9\n\tds_read_u8 %3, %9 offset:1\n\t"
                                             "ds_read_u8 %4, %10\n\tds_read_u8 %5, %10 offset:1\n\tds_read_u8 %6, %11\n\tds_read_u8 %7, %11 offset:1\n\t"
                                             "s_waitcnt lgkmcnt(0)"
                                             : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3]), "=&v"(t[4]), "=&v"(t[5]), "=&v"(t[6]), "=&v"(t[7])
                                             : "v"(tb + (uint32_t)a0), "v"(tb + (uint32_t)a1), "v"(tb + (uint32_t)a2), "v"(tb + (uint32_t)a3)
                                             : "memory");
                                float a255 = 1.0f;
                                done = shade_bytes<USE_TF, ERT>(t, u0 - f0, u1 - f1, u2 - f2, kexp, tf, acc[r], DIAG ? &a255 : nullptr);
                                if (DIAG) n_zero += a255 == 0.0f ? 1u : 0u;
                            } else {
                                uint32_t c[8];
                                c[0] = ld(a0); c[1] = ld(a0 + 4);
                                c[2] = ld(a1); c[3] = ld(a1 + 4);
                                c[4] = ld(a2); c[5] = ld(a2 + 4);
                                c[6] = ld(a3); c[7] = ld(a3 + 4);
                                float a255 = 1.0f;
                                done = shade<ALPHA_ONLY, USE_TF, ERT>(c, u0 - f0, u1 - f1, u2 - f2, kexp, tf, acc[r], DIAG ? &a255 : nullptr);
                                if (DIAG) n_zero += a255 == 0.0f ? 1u : 0u;
                            }
                        }
                        // (a ray that stopped keeps no use for dt: the sum may advance unconditionally)
                        dt[r] += VRM_STEP_SIZE;
                        lim = (done || dt[r] >= ray[r].t_max) ? -1 : lim;
                    }
                }
                active[r] = active[r] && lim >= 0;
            };
            // (Measured and dropped: a second copy of the interior loop without the per-step test `dt >= t_max`, taken when a
            // bound on the running sum shows that no ray of the wave can end in this chunk -- C3 -0.9 %, C4 +0.8 %.)
            // (two rays: one after the other through the same staged box)
            if (interior) {
                steps(std::true_type{}, std::integral_constant<int, 0>{});
                if constexpr (R > 1) steps(std::true_type{}, std::integral_constant<int, R - 1>{});
            } else {
                steps(std::false_type{}, std::integral_constant<int, 0>{});
                if constexpr (R > 1) steps(std::false_type{}, std::integral_constant<int, R - 1>{});
            }
        } else {
            // box too large for the LDS tile (far-away camera: a pixel spans many voxels): gather from global memory
#pragma unroll
            for (int r = 0; r < R; ++r)
            for (int j = 0; j < kk; ++j) {
                if (j < n_lane[r] && active[r]) {
                    if (DIAG) ++iters[r];
                    const float p0 = ray[r].ro[0] + ray[r].rd[0] * dt[r];
                    const float p1 = ray[r].ro[1] + ray[r].rd[1] * dt[r];
                    const float p2 = ray[r].ro[2] + ray[r].rd[2] * dt[r];
                    const bool inside = (p0 >= g.box_min[0]) & (p1 >= g.box_min[1]) & (p2 >= g.in_min_z) & (p0 < g.box_max[0]) &
                                        (p1 < g.box_max[1]) & (p2 < g.in_max_z);
                    bool done = false;
                    if (inside) {
                        if (DIAG) { ++n_inside[r]; ++n_exec[r]; }
                        // the same filter coordinate as the staged path: which path a sample takes depends on the chunking
                        float uu[3];
                        texel_coord(dt[r], uu, r);
                        const float u0 = uu[0], u1 = uu[1], u2 = uu[2];
                        const float f0 = floorf(u0), f1 = floorf(u1), f2 = floorf(u2);
                        uint32_t c[8];
                        if constexpr (PACKED) {
                            // the alpha byte is all the 1-channel filter reads -- through the SAME filter as the staged samples
                            // (shade_bytes): what a sample contributes must not depend on which path a chunking sends it down
                            fetch_global_alpha(g, (int)f0, (int)f1, (int)f2, c);
#pragma unroll
                            for (int k = 0; k < 8; ++k) c[k] >>= 24;
                            done = shade_bytes<USE_TF, ERT>(c, u0 - f0, u1 - f1, u2 - f2, kexp, tf, acc[r]);
                        } else {
                            fetch_global(g, (int)f0, (int)f1, (int)f2, c);
                            done = shade<ALPHA_ONLY, USE_TF, ERT>(c, u0 - f0, u1 - f1, u2 - f2, kexp, tf, acc[r]);
                        }
                    }
                    // (a ray that stopped keeps no use for dt: the sum may advance unconditionally)
                    dt[r] += VRM_STEP_SIZE;
                    active[r] = !(done || dt[r] >= ray[r].t_max);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            steps_left[r] -= n_lane[r]; // a lane that stopped early is inactive anyway
            active[r] = active[r] && steps_left[r] > 0;
        }
    }
    unsigned long long n_exec_lane = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        finish_colour<ALPHA_ONLY, USE_TF, ERT>(acc[r]);
        if (in_image[r]) store_pixel<DIAG>(v, P, px[r], py, acc[r], ray[r], iters[r], n_inside[r], n_exec[r]);
        n_exec_lane += n_exec[r];
    }
    if (DIAG && P.diag_counters) { // vrm_diag_read_counters: one atomic per counter and wave
        unsigned long long e = n_exec_lane, z = n_zero;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            e += __shfl_down(e, off, 64);
            z += __shfl_down(z, off, 64);
        }
        if (lane == 0) {
            atomicAdd(&P.diag_counters[0], e);
            atomicAdd(&P.diag_counters[1], z);
            atomicAdd(&P.diag_counters[2], (unsigned long long)st_chunks);
            atomicAdd(&P.diag_counters[3], (unsigned long long)st_staged);
            atomicAdd(&P.diag_counters[4], (unsigned long long)st_empty);
            atomicAdd(&P.diag_counters[5], (unsigned long long)st_probe_fail);
            atomicAdd(&P.diag_counters[6], (unsigned long long)st_fallback);
            atomicAdd(&P.diag_counters[7], (unsigned long long)st_retry);
        }
    }
}

// The march kernel: one wave per workgroup.  Two ways of handing out the tiles:
//  * one workgroup per tile and view (grid = tiles x views), dealt to the XCDs round-robin by the dispatcher, or
//  * P.queue != nullptr: as many workgroups as the chip holds, each taking tiles from its XCD's queue -- the same order,
//    (view, slot) = position q in the XCD's list, claimed with one atomic per tile -- and, once that queue is empty, from
//    the other XCDs' queues.  No workgroup waits behind another XCD's slow tiles (the hardware dispatcher hands workgroups
//    out in order) and the launch has no tail in which most of the chip idles.  Every wave leaves when all eight queues
//    are empty; a tile is marched exactly once whichever wave claims it, so the image does not depend on the mode.
template <bool ALPHA_ONLY, bool USE_TF, bool DIAG, bool SKIP, int CAP, bool PACKED, bool ERT, bool FAR, int RAYS = 1>
#ifdef VRM_AB_LDS_PAD
__global__ void __launch_bounds__(64, 2) march_lds_kernel
#else
// (waves per SIMD = what 160 KB of LDS per CU hold at CAP words per one-wave workgroup, which is also the register budget)
__global__ void __launch_bounds__(64, (163840 / (CAP * 4) / 4 > 8 ? 8 : 163840 / (CAP * 4) / 4 < 1 ? 1 : 163840 / (CAP * 4) / 4)) march_lds_kernel
#endif
(const VrmMarchParams P)
{
    if (P.queue == nullptr) {
        march_tile<ALPHA_ONLY, USE_TF, DIAG, SKIP, CAP, PACKED, ERT, FAR, RAYS>(P, P.views[blockIdx.y], blockIdx.x & 7u, blockIdx.x >> 3);
        return;
    }
    const uint32_t per_view = P.sblocks_x * P.sblocks_y * 8u; // tile slots of one view in one XCD's list
    const uint32_t total = per_view * P.n_views;
    // (Measured and dropped: claims of several tiles at once, +5 % for two and +40 % for eight -- what the L2 likes is
    // neighbouring tiles marched at the same time by neighbouring waves, not one after the other by one wave; a claim
    // issued one tile ahead to hide its latency, +5 % for the same reason, and 0 % when issued only in a tile's last chunks; counters kept in the XCD's own L2 without
    // helping other XCDs, which gives the whole gain back: the gain IS the balance between XCDs.)
    // Each list is claimed through VRM_QUEUE_SUBS counters, counter s handing out the tiles of the list's super-blocks s,
    // s + SUBS, ... one super-block after the other: atomics on one address complete one after the other, and four counters
    // are worth 1 % on C3 and 4 % on the packed march (more counters, or counters interleaved tile by tile instead of
    // super-block by super-block, lose it again: the waves on one counter should be enough to march a super-block's tiles
    // at about the same time).
    // The wave's place in the queues is ONE number that lives through the tiles (SGPRs are short): the counter it claims
    // from, xcd * SUBS + sub, in the low byte and the counters it has found exhausted above it.
    constexpr uint32_t NCTR = 8u * VRM_QUEUE_SUBS;
    static_assert((NCTR & (NCTR - 1u)) == 0u && NCTR <= 256u, "counter index wraps with a mask and fits a byte");
    uint32_t place = (blockIdx.x & 7u) * VRM_QUEUE_SUBS + (blockIdx.x >> 3) % VRM_QUEUE_SUBS;
    while ((place >> 8) < NCTR) {
        const uint32_t ctr = place & 0xffu, xcd = ctr / VRM_QUEUE_SUBS, sub = ctr % VRM_QUEUE_SUBS;
        uint32_t c = 0;
        if (threadIdx.x == 0) c = atomicAdd(&P.queue[ctr * 32u], 1u); // (counters 128 bytes apart)
        c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
        const uint32_t q = (((c >> 6) * VRM_QUEUE_SUBS + sub) << 6) + (c & 63u);
        if (q >= total) { // this counter's super-blocks are done: the next counter, after the last one the next XCD's list
            place = ((place & ~0xffu) + 0x100u) | ((ctr + 1u) & (NCTR - 1u));
            continue;
        }
        const uint32_t view = q / per_view;
        march_tile<ALPHA_ONLY, USE_TF, DIAG, SKIP, CAP, PACKED, ERT, FAR, RAYS>(P, P.views[view], xcd, q - view * per_view);
    }
}

// ---------------------------------------------------------------------------------------------
// Reference-equivalent sample count: the number of loop iterations raymarch() (wgsl:139-187) executes
// for every ray of a view when the early exit `alpha >= 1.0` does not fire (it cannot for
// a*attenuation*0.001 < ln 2).  Pure geometry: the same dt lattice, no sampling.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) count_samples_kernel(const VrmMarchParams P, unsigned long long *total)
{
    const VrmGridDev &g = P.grid;
    const uint32_t view = blockIdx.y;
    const uint32_t tile = blockIdx.x;
    const uint32_t ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t px = tx * TILE + (wave & 1u) * 8u + (lane & 7u);
    const uint32_t py = ty * TILE + (wave >> 1) * 8u + (lane >> 3);
    unsigned long long n = 0;
    if (px < P.width && py < P.height) {
        const Ray ray = setup_ray(P.views[view], g, px, py, P.width, P.height);
        if (ray.hit) {
            float dt = ray.t_min;
            for (int i = 0; i < VRM_MAX_STEP_AMOUNT; ++i) {
                ++n;
                dt += VRM_STEP_SIZE;
                if (dt >= ray.t_max) break;
            }
        }
    }
    // wave reduction, one atomic per wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_down(n, off, 64);
    if (lane == 0 && n) atomicAdd(total, n);
}

// ---------------------------------------------------------------------------------------------
// ingest kernels (HBM-bound, trivially parallel)
// ---------------------------------------------------------------------------------------------

// Rust `f32 as u8`: truncate toward zero, saturate, NaN -> 0 (loaders/netcdf.rs:54-57)
__device__ __forceinline__ uint32_t f32_as_u8(float v)
{
    v = v * 255.0f;
    if (!(v == v)) return 0u;
    v = fminf(fmaxf(v, 0.0f), 255.0f);
    return (uint32_t)v;
}

__global__ void __launch_bounds__(256) quantize_f32_kernel(const float *__restrict__ src, uint32_t channels, size_t n,
                                                           uint32_t *__restrict__ dst)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        uint32_t w;
        if (channels == 4) {
            w = f32_as_u8(src[i]) | (f32_as_u8(src[n + i]) << 8) | (f32_as_u8(src[2 * n + i]) << 16) |
                (f32_as_u8(src[3 * n + i]) << 24);
        } else {
            const uint32_t q = f32_as_u8(src[i]);
            w = q * 0x01010101u;
        }
        dst[i] = w;
    }
}

// loaders/dat.rs:62-74: big-endian combine, ((val / (2^(8n)-1)) * 255) as u8 (replicated to RGBA by the caller)
__device__ __forceinline__ uint32_t quantize_scalar(const uint8_t *__restrict__ raw, uint32_t bps, size_t i)
{
    float val, denom;
    if (bps == 2) {
        val = (float)(((uint32_t)raw[2 * i] << 8) | raw[2 * i + 1]);
        denom = 65535.0f;
    } else {
        val = (float)raw[i];
        denom = 255.0f;
    }
    float q = (val / denom) * 255.0f;
    q = fminf(fmaxf(q, 0.0f), 255.0f);
    return (uint32_t)q;
}

__global__ void __launch_bounds__(256) quantize_scalar_kernel(const uint8_t *__restrict__ raw, uint32_t bps, size_t n,
                                                              uint32_t *__restrict__ dst)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = quantize_scalar(raw, bps, i) * 0x01010101u;
}

// The alpha of voxel i as the two builders below read it: from the x-fastest RGBA8 words (BPS = 0), or straight from the raw
// samples of a scalar volume (BPS = 1, 2: the direct ingest, which never holds the RGBA form -- the value every channel of
// that voxel would get, quantised as dat.rs does).
template <int BPS>
__device__ __forceinline__ uint32_t source_alpha(const void *__restrict__ src, size_t i)
{
    if constexpr (BPS == 0) return static_cast<const uint32_t *>(src)[i] >> 24;
    else return quantize_scalar(static_cast<const uint8_t *>(src), (uint32_t)BPS, i);
}

// max alpha of every 4x4x4-voxel cell: the look-up behind exact empty-space skipping.  One wave per 8x8x8-voxel
// region = 2x2x2 cells; lane (lx, ly) walks z, then the 4x4 lanes of a cell column are reduced by xor shuffles.
template <int BPS>
__global__ void __launch_bounds__(64) build_occupancy_kernel(const void *__restrict__ vox, uint32_t X, uint32_t Y, uint32_t Z,
                                                             uint32_t rx_n, uint32_t ry_n, uint32_t ox_n, uint32_t oy_n,
                                                             uint32_t oz_n, uint8_t *__restrict__ occ)
{
    const uint32_t reg = blockIdx.x;
    const uint32_t rz = reg / (rx_n * ry_n), r = reg - rz * rx_n * ry_n, ry = r / rx_n, rx = r - ry * rx_n;
    const uint32_t lx = threadIdx.x & 7u, ly = threadIdx.x >> 3;
    const uint32_t x = rx * 8u + lx, y = ry * 8u + ly;
    uint32_t m[2] = {0, 0};
    if (x < X && y < Y)
        for (uint32_t k = 0; k < 8u; ++k) {
            const uint32_t z = rz * 8u + k;
            if (z < Z) m[k >> 2] = max(m[k >> 2], source_alpha<BPS>(vox, ((size_t)z * Y + y) * X + x));
        }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        uint32_t v = m[h];
        v = max(v, (uint32_t)__shfl_xor((int)v, 1, 64));
        v = max(v, (uint32_t)__shfl_xor((int)v, 2, 64));
        v = max(v, (uint32_t)__shfl_xor((int)v, 8, 64));
        v = max(v, (uint32_t)__shfl_xor((int)v, 16, 64));
        const uint32_t cx = rx * 2u + (lx >> 2), cy = ry * 2u + (ly >> 2), cz = rz * 2u + (uint32_t)h;
        if ((lx & 3u) == 0 && (ly & 3u) == 0 && cx < ox_n && cy < oy_n && cz < oz_n) occ[((size_t)cz * oy_n + cy) * ox_n + cx] = (uint8_t)v;
    }
}

// one level of the occupancy max pyramid from the level below: dst cell = max of the (up to) 2x2x2 src cells it covers
__global__ void __launch_bounds__(256) build_occupancy_mip_kernel(const uint8_t *__restrict__ src, uint32_t sx, uint32_t sy, uint32_t sz,
                                                                 uint8_t *__restrict__ dst, uint32_t dx, uint32_t dy, uint32_t dz)
{
    const size_t n = (size_t)dx * dy * dz, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t z = (uint32_t)(i / ((size_t)dx * dy)), r = (uint32_t)(i - (size_t)z * dx * dy), y = r / dx, x = r - y * dx;
        uint32_t m = 0;
        for (uint32_t k = 0; k < 2u; ++k)
            for (uint32_t j = 0; j < 2u; ++j)
                for (uint32_t l = 0; l < 2u; ++l) {
                    const uint32_t cx = min(2u * x + l, sx - 1u), cy = min(2u * y + j, sy - 1u), cz = min(2u * z + k, sz - 1u);
                    m = max(m, (uint32_t)src[((size_t)cz * sy + cy) * sx + cx]);
                }
        dst[i] = (uint8_t)m;
    }
}

// x-fastest voxels -> 4x4x2-texel bricks (VrmGridDev::bricks).  One thread per 16-byte quad of a brick, output
// fully coalesced; texels beyond the volume (dims not multiples of 4,4,2) replicate the edge and are never sampled.
__global__ void __launch_bounds__(256) build_bricks_kernel(const uint32_t *__restrict__ vox, uint32_t X, uint32_t Y, uint32_t Z,
                                                           uint32_t nbx, uint32_t nby, size_t n_quads, u32x4 *__restrict__ bricks)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += stride) {
        const size_t brick = i >> 3;
        const uint32_t local = (uint32_t)i & 7u;
        const size_t bz = brick / ((size_t)nbx * nby), r = brick - bz * nbx * nby;
        const uint32_t by = (uint32_t)(r / nbx), bx = (uint32_t)(r - (size_t)by * nbx);
        const uint32_t y = min(by * 4u + (local & 3u), Y - 1u), z = min((uint32_t)bz * 2u + (local >> 2), Z - 1u);
        const uint32_t *row = vox + ((size_t)z * Y + y) * X;
        const uint32_t x = bx * 4u;
        u32x4 w;
        w.x = row[min(x, X - 1u)];
        w.y = row[min(x + 1u, X - 1u)];
        w.z = row[min(x + 2u, X - 1u)];
        w.w = row[min(x + 3u, X - 1u)];
        bricks[i] = w;
    }
}

// The alpha channel alone, bricked like `bricks` over words of four x-adjacent texels (VrmGridDev::abricks): one
// 16-byte quad = 16 texels of one (y, z) row, edge-replicated past the volume.
template <int BPS>
__global__ void __launch_bounds__(256) build_alpha_bricks_kernel(const void *__restrict__ vox, uint32_t X, uint32_t Y, uint32_t Z,
                                                                 uint32_t nbxa, uint32_t nby, size_t n_quads,
                                                                 u32x4 *__restrict__ abricks)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_quads; i += stride) {
        const size_t brick = i >> 3;
        const uint32_t local = (uint32_t)i & 7u;
        const size_t bz = brick / ((size_t)nbxa * nby), r = brick - bz * nbxa * nby;
        const uint32_t by = (uint32_t)(r / nbxa), bx = (uint32_t)(r - (size_t)by * nbxa);
        const uint32_t y = min(by * 4u + (local & 3u), Y - 1u), z = min((uint32_t)bz * 2u + (local >> 2), Z - 1u);
        const size_t row = ((size_t)z * Y + y) * X;
        const uint32_t x = bx * 16u;
        uint32_t w[4];
#pragma unroll
        for (uint32_t q = 0; q < 4u; ++q) {
            w[q] = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) w[q] |= source_alpha<BPS>(vox, row + min(x + 4u * q + k, X - 1u)) << (8u * k);
        }
        u32x4 o;
        o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
        abricks[i] = o;
    }
}

// The x-fastest RGBA8 voxels of a SCALAR volume back from its packed alpha bricks ([b,b,b,b], dat.rs:74): what a scalar
// volume is expanded with when something needs the 4 B/voxel form again (download, broadcast, the RGBA-staged kernels).
__global__ void __launch_bounds__(256) expand_alpha_bricks_kernel(const uint32_t *__restrict__ abricks, uint32_t X, uint32_t Y, uint32_t Z,
                                                                  uint32_t nbxa, uint32_t nby, uint32_t *__restrict__ vox)
{
    const size_t n = (size_t)X * Y * Z, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t z = (uint32_t)(i / ((size_t)X * Y)), r = (uint32_t)(i - (size_t)z * X * Y), y = r / X, x = r - y * X;
        const size_t w = ((((size_t)(z >> 1) * nby + (y >> 2)) * nbxa + (x >> 4)) << 5) + ((z & 1u) << 4) + ((y & 3u) << 2) + ((x >> 2) & 3u);
        vox[i] = ((abricks[w] >> (8u * (x & 3u))) & 0xffu) * 0x01010101u;
    }
}

// ---------------------------------------------------------------------------------------------
// Sort-last compositing: `n_slabs` premultiplied f32 RGBA partials of the same image rows, slab 0
// lowest in z, are blended with the associative "over" operator in the order the ray visits the slabs -- ascending
// when rd.z >= 0, descending otherwise (rd.z as fs_main forms it, wgsl:98-108; its sign is that of the
// un-normalised z) -- and stored like the single-GPU path stores its pixel.  HBM-streaming.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) composite_kernel(const float4 *__restrict__ partials, uint32_t n_slabs, VrmCompositeView cv,
                                                        uint32_t width, uint32_t height, uint32_t row0, uint32_t rows,
                                                        uint32_t *__restrict__ out_rgba8, float4 *__restrict__ out_f32)
{
    const size_t n = (size_t)rows * width, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t yb = (uint32_t)(i / width), px = (uint32_t)(i - (size_t)yb * width), py = row0 + yb;
        const float uvx = ((float)px + 0.5f) / (float)width;
        const float uvy = 1.0f - ((float)py + 0.5f) / (float)height;
        const float sx = uvx * 2.0f - 1.0f, sy = uvy * 2.0f - 1.0f;
        const float wz = ((cv.ivp[2] * sx + cv.ivp[6] * sy) + cv.ivp[10] * 1.0f) + cv.ivp[14] * 1.0f;
        const bool ascending = wz >= 0.0f;
        Accum acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (uint32_t s = 0; s < n_slabs; ++s) {
            const uint32_t k = ascending ? s : n_slabs - 1u - s;
            const float4 p = partials[(size_t)k * n + i];
            const float om = 1.0f - acc.alpha;
            acc.cr = __builtin_fmaf(om, p.x, acc.cr);
            acc.cg = __builtin_fmaf(om, p.y, acc.cg);
            acc.cb = __builtin_fmaf(om, p.z, acc.cb);
            acc.alpha = __builtin_fmaf(om, p.w, acc.alpha);
        }
        if (out_rgba8) out_rgba8[i] = unorm8(acc.cr) | (unorm8(acc.cg) << 8) | (unorm8(acc.cb) << 16) | (unorm8(acc.alpha) << 24);
        if (out_f32) out_f32[i] = make_float4(acc.cr, acc.cg, acc.cb, acc.alpha);
    }
}

// ---------------------------------------------------------------------------------------------
// evaluation reductions, HBM-streaming
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_sum_store(double v, double *partial)
{
    __shared__ double red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// compare_to_netcdf_rmse (compare/netcdf.rs:17-69): per voxel (alpha - data3)^2, and unless density_only the three
// transfer-function colours of alpha against data0..2; every term in f32 exactly as the reference forms it, the
// sum in f64 per block (the reference's single sequential f32 sum is neither parallel nor accurate at 10^8 terms).
__global__ void __launch_bounds__(256) compare_volume_kernel(const uint32_t *__restrict__ vox, const float *__restrict__ data,
                                                             size_t n, int density_only, const float ca0, const float ca1,
                                                             const float ca2, const float cb0, const float cb1, const float cb2,
                                                             const float cc0, const float cc1, const float cc2,
                                                             double *__restrict__ partial)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float alpha = (float)(vox[i] >> 24) / 255.0f;                  // netcdf.rs:31
        const float d3 = alpha - data[3 * n + i];
        float t = d3 * d3;                                                   // netcdf.rs:32
        acc += (double)t;
        if (!density_only) {
            const bool lo = alpha < 0.5f;                                    // netcdf.rs:41-57
            const float s0 = lo ? ca0 : cb0, s1 = lo ? ca1 : cb1, s2 = lo ? ca2 : cb2;
            const float e0 = lo ? cb0 : cc0, e1 = lo ? cb1 : cc1, e2 = lo ? cb2 : cc2;
            // remap(value, min1, max1, 0, 1) = 0 + (value - min1) * (1 - 0) / (max1 - min1)   (netcdf.rs:81-83)
            const float a = lo ? 0.0f + (alpha - 0.0f) * (1.0f - 0.0f) / (0.5f - 0.0f) : 0.0f + (alpha - 0.5f) * (1.0f - 0.0f) / (1.0f - 0.5f);
            const float r = s0 * (1.0f - a) + e0 * a, gcol = s1 * (1.0f - a) + e1 * a, b = s2 * (1.0f - a) + e2 * a; // mix, :77-79
            const float dr = r - data[i], dg = gcol - data[n + i], db = b - data[2 * n + i];
            acc += (double)(dr * dr);
            acc += (double)(dg * dg);
            acc += (double)(db * db);
        }
    }
    block_sum_store(acc, partial);
}

// skimage.metrics.mean_squared_error on uint8 images (similarity.py:59-61): the exact integer sum of squared
// differences; rows of row_bytes (a multiple of 4) bytes, pitches in bytes (multiples of 4).
__global__ void __launch_bounds__(256) ssd_u8_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, size_t row_words,
                                                     size_t rows, size_t pitch_a, size_t pitch_b,
                                                     unsigned long long *__restrict__ partial)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x, n = row_words * rows;
    unsigned long long acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const size_t row = i / row_words, col = i - row * row_words;
        const uint32_t wa = *reinterpret_cast<const uint32_t *>(a + row * pitch_a + 4 * col);
        const uint32_t wb = *reinterpret_cast<const uint32_t *>(b + row * pitch_b + 4 * col);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = (int)((wa >> (8 * k)) & 0xffu) - (int)((wb >> (8 * k)) & 0xffu);
            acc += (unsigned long long)(d * d);
        }
    }
    __shared__ unsigned long long red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63u) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// skimage.metrics.structural_similarity as similarity.py:65 calls it (uint8 images, channel_axis=-1, defaults
// otherwise): 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample covariance (49/48), f64 arithmetic, the 3-pixel
// border cropped before the mean -- so only whole windows count and the filter's edge mode never shows. One item =
// one (interior pixel, channel); the five window sums are exact integers (skimage's are f64 running means of the
// same integers), the SSIM term is formed in f64 and summed per block.
__global__ void __launch_bounds__(256) ssim_u8_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, uint32_t width,
                                                      uint32_t height, uint32_t channels, double c1, double c2,
                                                      double *__restrict__ partial)
{
    const uint32_t iw = width - 6u, ih = height - 6u;
    const size_t n = (size_t)iw * ih * channels, stride = (size_t)gridDim.x * blockDim.x;
    const size_t pitch = (size_t)width * channels;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t c = (uint32_t)(i % channels);
        const size_t px = i / channels;
        const uint32_t x = (uint32_t)(px % iw), y = (uint32_t)(px / iw);
        uint32_t sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (uint32_t dy = 0; dy < 7u; ++dy) {
            const size_t row = (size_t)(y + dy) * pitch + (size_t)x * channels + c;
#pragma unroll
            for (uint32_t dx = 0; dx < 7u; ++dx) {
                const uint32_t va = a[row + dx * channels], vb = b[row + dx * channels];
                sx += va;
                sy += vb;
                sxx += va * va;
                syy += vb * vb;
                sxy += va * vb;
            }
        }
        const double ux = (double)sx / 49.0, uy = (double)sy / 49.0;
        const double uxx = (double)sxx / 49.0, uyy = (double)syy / 49.0, uxy = (double)sxy / 49.0;
        const double cov_norm = 49.0 / 48.0;
        const double vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
        const double a1 = 2.0 * ux * uy + c1, a2 = 2.0 * vxy + c2, b1 = ux * ux + uy * uy + c1, b2 = vx + vy + c2;
        acc += (a1 * a2) / (b1 * b2);
    }
    block_sum_store(acc, partial);
}

__global__ void __launch_bounds__(256) fill_kernel(uint32_t *dst, size_t n, uint32_t value)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = value;
}

__global__ void __launch_bounds__(256) detect_nonscalar_kernel(const uint32_t *__restrict__ vox, size_t n, uint32_t *flag)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t w = vox[i];
        bad |= (w != (w >> 24) * 0x01010101u);
    }
    if (__any(bad)) {
        if ((threadIdx.x & 63u) == 0) atomicOr(flag, 1u);
    }
}

#ifdef VRM_ISA_ONE
// ISA inspection build (tools/isa_blocks.py --one): a single instantiation of the march kernel, no launchers -- compiles in
// seconds instead of minutes.  VRM_ISA_ONE = the template arguments.
template __global__ void march_lds_kernel<VRM_ISA_ONE>(const VrmMarchParams P);
} // namespace
#else
inline uint32_t stream_blocks(size_t n)
{
    size_t b = (n + 255) / 256;
    const size_t cap = 256 * 16; // 256 CUs x 16 blocks keeps every CU streaming
    return (uint32_t)(b < cap ? (b ? b : 1) : cap);
}

inline uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
// one-wave workgroups the current device holds at once with `lds_bytes` of LDS each (160 KB per CU on gfx950)
inline uint32_t resident_waves(uint32_t lds_bytes)
{
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return (uint32_t)cus * (160u * 1024u / lds_bytes);
}

template <bool A, bool T, bool D, bool E>
void launch_variant(VrmMarchParams p, uint32_t n_views, int variant, bool skip, hipStream_t stream)
{
    p.n_views = n_views;
    if (variant == 0) {
        const dim3 grid(p.tiles_x * p.tiles_y, n_views, 1), block(256, 1, 1);
        hipLaunchKernelGGL((march_linear_kernel<A, T, D>), grid, block, 0, stream, p);
    } else {
        const dim3 block(64, 1, 1);
        // texels a ray advances per step (the step is 0.001 world units whatever the resolution, wgsl:126)
        float tps = 0.0f;
        for (int a = 0; a < 3; ++a) tps = fmaxf(tps, VRM_STEP_SIZE * p.grid.tex_scale[a]);
#define VRM_LDS(CAPV, DEPTH, PK, NR)                                                                              \
    do {                                                                                                          \
        /* two rays per lane: wave tiles of 16x8 pixels, super-blocks of 128x64 */                                 \
        if ((NR) == 2) p.sblocks_x = (((p.width + 127u) / 128u) + 7u) & ~7u;                                       \
        const dim3 grid(p.sblocks_x * p.sblocks_y * 64u, n_views, 1);                                              \
        /* chunk length: about DEPTH texels of depth per chunk unless the caller fixed it */                      \
        if (p.chunk_steps == 0) p.chunk_steps = (uint32_t)fminf(fmaxf(floorf((DEPTH) / tps + 0.5f), 2.0f), 64.0f); \
        /* tile queues: as many one-wave workgroups as the chip holds at CAPV words of LDS each.  Variant 1 chooses:     \
         * queues balance the XCDs (C3 -7 %, small images up to -44 %) but expose one atomic round trip per tile, which   \
         * costs the packed march of large images more than it gains (+6 % at 1024^2, +9 % at 1536^2). */               \
        if (variant == 1 && (PK) && skip && p.sblocks_x * p.sblocks_y * 64u > 12288u) p.queue = nullptr;                \
        /* ... and a launch of one or two views is over before the balance matters (1 view: 0.74 vs 0.79 ms at 1024^2) */ \
        if (variant == 1 && n_views < 3u) p.queue = nullptr;                                                       \
        /* (queue positions are 32-bit: a launch of 2^32 tiles or more, which no image memory holds, keeps the grid form) */ \
        if ((uint64_t)p.sblocks_x * p.sblocks_y * 64u * n_views >= (1ull << 32)) p.queue = nullptr;                \
        const dim3 gridq(p.queue ? min_u32(p.tiles_total(n_views), resident_waves((CAPV) * 4u)) : grid.x, p.queue ? 1u : grid.y, 1); \
        if (p.queue) (void)hipMemsetAsync(p.queue, 0, VRM_QUEUE_BYTES, stream);                                               \
        if (skip) hipLaunchKernelGGL((march_lds_kernel<A, T, D, true, CAPV, PK, E, false, NR>), gridq, block, 0, stream, p);   \
        else hipLaunchKernelGGL((march_lds_kernel<A, T, D, false, CAPV, PK, E, false, NR>), gridq, block, 0, stream, p);       \
        /* views of far-away cameras are left to the FAR instance (its waves take only those; one ray per lane: the   \
         * host asks for two rays only when no view of the batch is far) */                                        \
        if (p.any_far && (NR) == 1) {                                                                             \
            if (p.queue) (void)hipMemsetAsync(p.queue, 0, VRM_QUEUE_BYTES, stream);                                           \
            if (skip) hipLaunchKernelGGL((march_lds_kernel<A, T, D, true, CAPV, PK, true, true>), gridq, block, 0, stream, p);  \
            else hipLaunchKernelGGL((march_lds_kernel<A, T, D, false, CAPV, PK, true, true>), gridq, block, 0, stream, p);      \
        }                                                                                                         \
    } while (0)
        // (LDS words per wave, box depth in texels): LDS per wave sets the waves per CU (160 KB / CU).  10 KB (16 waves per
        // CU) is the measured optimum for 256^3 .. 1024^3 volumes; variant 2 keeps the 8 KB tile (20 waves per CU, 5 per
        // SIMD) selectable for A/B runs (VRM_OPT_KERNEL): more resident waves do not pay for the shallower chunks.
        if (variant == 2) {
            VRM_LDS(2048, 6.0f, false, 1);
        } else {
            // 1-channel march (scalar volume or transfer function): staged from the packed alpha bricks -- a quarter of
            // the bytes and of the tile's x extent, so 15 texels of depth pay where the RGBA tile holds 8 (measured optima:
            // 15 steps at 1024^3, 27 at 512^3 + transfer function, flat at 256^3).
            // VRM_OPT_PACKED_ALPHA = 0 keeps the RGBA source.
            bool done = false;
            if constexpr (A) {
                if (p.grid.abricks != nullptr && p.packed_alpha != 0) {
                    // two rays per lane where the host found every view's pixels narrower than ~0.6 texel (scalar volumes without
                    // the transfer function: that instance has no registers to spare); never for far or slab-partitioned views
                    bool two = false;
                    if constexpr (!T) two = p.rays == 2u && !p.any_far && !p.grid.slab;
                    if (two) {
                        if constexpr (!T) VRM_LDS(VRM_CAP_PACKED, VRM_RAYS2_DEPTH, true, 2);
                    } else {
                        VRM_LDS(VRM_CAP_PACKED, 15.0f, true, 1);
                    }
                    done = true;
                }
            }
            if (!done) VRM_LDS(VRM_CAP_DEFAULT, 8.0f, false, 1); // 10 KB -> 16 waves/CU (variant 1, the default)
        }
#undef VRM_LDS
    }
}

template <bool A, bool T, bool D>
void launch_ert(const VrmMarchParams &p, uint32_t n_views, int variant, bool skip, hipStream_t stream)
{
    // The early exit `alpha >= 1.0` (wgsl:176) can only fire if some sample has alpha_src >= 0.5, i.e.
    // a * attenuation * 0.001 >= ln 2 with a <= 1: attenuation >= 693 (the GUI's range ends at 100).
    // Below 690 the kernels are instantiated without the test; anything else (NaN included) keeps it.
    if (p.attenuation < 690.0f) launch_variant<A, T, D, false>(p, n_views, variant, skip, stream);
    else launch_variant<A, T, D, true>(p, n_views, variant, skip, stream);
}

} // namespace

hipError_t vrm_launch_build_bricks(const uint32_t *vox, const uint32_t dim[3], uint32_t *bricks, hipStream_t stream)
{
    const uint32_t nbx = (dim[0] + 3) / 4, nby = (dim[1] + 3) / 4, nbz = (dim[2] + 1) / 2;
    const size_t n_quads = (size_t)nbx * nby * nbz * 8;
    hipLaunchKernelGGL(build_bricks_kernel, dim3(stream_blocks(n_quads)), dim3(256), 0, stream, vox, dim[0], dim[1], dim[2], nbx, nby,
                       n_quads, reinterpret_cast<u32x4 *>(bricks));
    return hipGetLastError();
}

size_t vrm_alpha_bricks_bytes(const uint32_t dim[3])
{
    return (size_t)((dim[0] + 15) / 16) * ((dim[1] + 3) / 4) * ((dim[2] + 1) / 2) * 128;
}

hipError_t vrm_launch_build_alpha_bricks(const void *src, uint32_t bps, const uint32_t dim[3], uint32_t *abricks, hipStream_t stream)
{
    const uint32_t nbxa = (dim[0] + 15) / 16, nby = (dim[1] + 3) / 4, nbz = (dim[2] + 1) / 2;
    const size_t n_quads = (size_t)nbxa * nby * nbz * 8;
    const dim3 grid(stream_blocks(n_quads)), block(256);
    u32x4 *out = reinterpret_cast<u32x4 *>(abricks);
    if (bps == 0) hipLaunchKernelGGL(build_alpha_bricks_kernel<0>, grid, block, 0, stream, src, dim[0], dim[1], dim[2], nbxa, nby, n_quads, out);
    else if (bps == 1) hipLaunchKernelGGL(build_alpha_bricks_kernel<1>, grid, block, 0, stream, src, dim[0], dim[1], dim[2], nbxa, nby, n_quads, out);
    else hipLaunchKernelGGL(build_alpha_bricks_kernel<2>, grid, block, 0, stream, src, dim[0], dim[1], dim[2], nbxa, nby, n_quads, out);
    return hipGetLastError();
}

hipError_t vrm_launch_expand_alpha_bricks(const uint32_t *abricks, const uint32_t dim[3], uint32_t *vox, hipStream_t stream)
{
    const uint32_t nbxa = (dim[0] + 15) / 16, nby = (dim[1] + 3) / 4;
    hipLaunchKernelGGL(expand_alpha_bricks_kernel, dim3(stream_blocks((size_t)dim[0] * dim[1] * dim[2])), dim3(256), 0, stream, abricks, dim[0],
                       dim[1], dim[2], nbxa, nby, vox);
    return hipGetLastError();
}

size_t vrm_occupancy_bytes(const uint32_t dim[3], uint32_t off[VRM_OCC_LEVELS])
{
    size_t total = 0;
    for (int l = 0; l < VRM_OCC_LEVELS; ++l) {
        size_t n = 1;
        for (int a = 0; a < 3; ++a) n *= (size_t)((((dim[a] + 3) / 4) + (1u << l) - 1u) >> l);
        off[l] = (uint32_t)total;      // level 0 of a 2^40-texel volume is 2^34 bytes: the coarser levels start beyond 4 GiB there,
        total += (n + 255) & ~(size_t)255; // which the march kernel checks before using them (occ_off[l] == 0 for l > 0 => level unused)
    }
    if (total > 0xffffffffull)
        for (int l = 1; l < VRM_OCC_LEVELS; ++l) off[l] = 0;
    return total;
}

hipError_t vrm_launch_build_occupancy(const void *src, uint32_t bps, const uint32_t dim[3], uint8_t *occ, hipStream_t stream)
{
    const uint32_t rx = (dim[0] + 7) / 8, ry = (dim[1] + 7) / 8, rz = (dim[2] + 7) / 8;
    const uint32_t od[3] = {(dim[0] + 3) / 4, (dim[1] + 3) / 4, (dim[2] + 3) / 4};
    const dim3 grid(rx * ry * rz), block(64);
    if (bps == 0) hipLaunchKernelGGL(build_occupancy_kernel<0>, grid, block, 0, stream, src, dim[0], dim[1], dim[2], rx, ry, od[0], od[1], od[2], occ);
    else if (bps == 1) hipLaunchKernelGGL(build_occupancy_kernel<1>, grid, block, 0, stream, src, dim[0], dim[1], dim[2], rx, ry, od[0], od[1], od[2], occ);
    else hipLaunchKernelGGL(build_occupancy_kernel<2>, grid, block, 0, stream, src, dim[0], dim[1], dim[2], rx, ry, od[0], od[1], od[2], occ);
    uint32_t off[VRM_OCC_LEVELS];
    (void)vrm_occupancy_bytes(dim, off);
    for (int l = 1; l < VRM_OCC_LEVELS && off[l] != 0; ++l) {
        uint32_t s[3], d[3];
        for (int a = 0; a < 3; ++a) {
            s[a] = (od[a] + (1u << (l - 1)) - 1u) >> (l - 1);
            d[a] = (od[a] + (1u << l) - 1u) >> l;
        }
        hipLaunchKernelGGL(build_occupancy_mip_kernel, dim3(stream_blocks((size_t)d[0] * d[1] * d[2])), dim3(256), 0, stream,
                           occ + off[l - 1], s[0], s[1], s[2], occ + off[l], d[0], d[1], d[2]);
    }
    return hipGetLastError();
}

hipError_t vrm_launch_march(const VrmMarchParams &p, uint32_t n_views, int variant, bool diagnostics, bool skip, hipStream_t stream)
{
    const bool tf = p.use_tf != 0;
    const bool alpha_only = p.grid.scalar != 0;
    if (diagnostics) {
        if (tf) launch_ert<true, true, true>(p, n_views, variant, skip, stream);
        else if (alpha_only) launch_ert<true, false, true>(p, n_views, variant, skip, stream);
        else launch_ert<false, false, true>(p, n_views, variant, skip, stream);
    } else {
        if (tf) launch_ert<true, true, false>(p, n_views, variant, skip, stream);
        else if (alpha_only) launch_ert<true, false, false>(p, n_views, variant, skip, stream);
        else launch_ert<false, false, false>(p, n_views, variant, skip, stream);
    }
    return hipGetLastError();
}

hipError_t vrm_launch_count_samples(const VrmMarchParams &p, uint32_t n_views, unsigned long long *total_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(count_samples_kernel, dim3(p.tiles_x * p.tiles_y, n_views, 1), dim3(256, 1, 1), 0, stream, p, total_dev);
    return hipGetLastError();
}

hipError_t vrm_launch_quantize_f32(const float *src, uint32_t channels, size_t n, uint32_t *dst, hipStream_t stream)
{
    hipLaunchKernelGGL(quantize_f32_kernel, dim3(stream_blocks(n)), dim3(256), 0, stream, src, channels, n, dst);
    return hipGetLastError();
}

hipError_t vrm_launch_quantize_scalar(const uint8_t *raw, uint32_t bps, size_t n, uint32_t *dst, hipStream_t stream)
{
    hipLaunchKernelGGL(quantize_scalar_kernel, dim3(stream_blocks(n)), dim3(256), 0, stream, raw, bps, n, dst);
    return hipGetLastError();
}

hipError_t vrm_launch_fill(uint32_t *dst, size_t n, uint32_t value, hipStream_t stream)
{
    hipLaunchKernelGGL(fill_kernel, dim3(stream_blocks(n)), dim3(256), 0, stream, dst, n, value);
    return hipGetLastError();
}

hipError_t vrm_launch_detect_nonscalar(const uint32_t *vox, size_t n, uint32_t *flag_dev, hipStream_t stream)
{
    hipLaunchKernelGGL(detect_nonscalar_kernel, dim3(stream_blocks(n)), dim3(256), 0, stream, vox, n, flag_dev);
    return hipGetLastError();
}

uint32_t vrm_reduce_blocks(size_t n_items) { return stream_blocks(n_items); }

hipError_t vrm_launch_compare_volume(const uint32_t *vox, const float *data_czyx, size_t n, bool density_only, const float tf9[9],
                                     double *partial, uint32_t blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(compare_volume_kernel, dim3(blocks), dim3(256), 0, stream, vox, data_czyx, n, density_only ? 1 : 0, tf9[0],
                       tf9[1], tf9[2], tf9[3], tf9[4], tf9[5], tf9[6], tf9[7], tf9[8], partial);
    return hipGetLastError();
}

hipError_t vrm_launch_ssd_u8(const uint8_t *a, const uint8_t *b, size_t row_words, size_t rows, size_t pitch_a, size_t pitch_b,
                             unsigned long long *partial, uint32_t blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(ssd_u8_kernel, dim3(blocks), dim3(256), 0, stream, a, b, row_words, rows, pitch_a, pitch_b, partial);
    return hipGetLastError();
}

hipError_t vrm_launch_ssim_u8(const uint8_t *a, const uint8_t *b, uint32_t width, uint32_t height, uint32_t channels, double c1,
                              double c2, double *partial, uint32_t blocks, hipStream_t stream)
{
    hipLaunchKernelGGL(ssim_u8_kernel, dim3(blocks), dim3(256), 0, stream, a, b, width, height, channels, c1, c2, partial);
    return hipGetLastError();
}

hipError_t vrm_launch_composite(const float *partials, uint32_t n_slabs, const VrmCompositeView &cv, uint32_t width, uint32_t height,
                                uint32_t row0, uint32_t rows, uint32_t *out_rgba8, float *out_f32, hipStream_t stream)
{
    hipLaunchKernelGGL(composite_kernel, dim3(stream_blocks((size_t)rows * width)), dim3(256), 0, stream,
                       reinterpret_cast<const float4 *>(partials), n_slabs, cv, width, height, row0, rows, out_rgba8,
                       reinterpret_cast<float4 *>(out_f32));
    return hipGetLastError();
}
#endif // VRM_ISA_ONE
