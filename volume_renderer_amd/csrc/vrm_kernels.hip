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
#include "vrm_device.h"

#pragma clang fp contract(off)

namespace {

constexpr int TILE = 16;        // workgroup = 16x16 pixels = 4 waves, each wave an 8x8 pixel tile
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
        r.rd[k] = ((w3[k]) / (len));
        r.ro[k] = v.pos[k];
    }
    float t_min = ((g.box_min[0] - r.ro[0]) / (r.rd[0]));
    float t_max = ((g.box_max[0] - r.ro[0]) / (r.rd[0]));
    if (t_min > t_max) { float t = t_min; t_min = t_max; t_max = t; }
    float t_y_min = ((g.box_min[1] - r.ro[1]) / (r.rd[1]));
    float t_y_max = ((g.box_max[1] - r.ro[1]) / (r.rd[1]));
    if (t_y_min > t_y_max) { float t = t_y_min; t_y_min = t_y_max; t_y_max = t; }
    bool hit = !((t_min > t_y_max) || (t_y_min > t_max));
    if (t_y_min > t_min) t_min = t_y_min;
    if (t_y_max < t_max) t_max = t_y_max;
    float t_z_min = ((g.box_min[2] - r.ro[2]) / (r.rd[2]));
    float t_z_max = ((g.box_max[2] - r.ro[2]) / (r.rd[2]));
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

__device__ __forceinline__ float ub(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xffu); }

// ---------------------------------------------------------------------------------------------
// Variant 0: linear x-fastest RGBA8 volume, software trilinear straight from global memory.
// One thread per pixel, one wave per 8x8 pixel tile.  Correctness baseline.
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

    float cr = 0.0f, cg = 0.0f, cb = 0.0f, alpha = 0.0f; // colour accumulators in 0..255 texel units when !USE_TF
    uint32_t iters = 0, n_inside = 0;

    if (ray.hit) {
        // alpha_src = 1 - exp(-a*step*att) = 1 - exp2(a255 * kexp), a255 = filtered alpha in 0..255
        const float kexp = -(VRM_STEP_SIZE * P.attenuation) * LOG2E * (1.0f / 255.0f);
        const uint32_t X = g.dim[0], Y = g.dim[1];
        const int mx = (int)g.dim[0] - 1, my = (int)g.dim[1] - 1, mz = (int)g.dim[2] - 1;
        float dt = ray.t_min;
        for (int i = 0; i < VRM_MAX_STEP_AMOUNT; ++i) {
            ++iters;
            // exact geometry: p = ro + rd*dt, separately rounded (wgsl:141)
            const float p0 = ray.ro[0] + ray.rd[0] * dt;
            const float p1 = ray.ro[1] + ray.rd[1] * dt;
            const float p2 = ray.ro[2] + ray.rd[2] * dt;
            const bool inside = (p0 >= g.box_min[0]) & (p1 >= g.box_min[1]) & (p2 >= g.box_min[2]) & (p0 < g.box_max[0]) &
                                (p1 < g.box_max[1]) & (p2 < g.box_max[2]); // wgsl:218-223
            if (inside) {
                if (DIAG) ++n_inside;
                const float u0 = __builtin_fmaf(p0, g.tex_scale[0], g.tex_bias[0]);
                const float u1 = __builtin_fmaf(p1, g.tex_scale[1], g.tex_bias[1]);
                const float u2 = __builtin_fmaf(p2, g.tex_scale[2], g.tex_bias[2]);
                const float f0 = floorf(u0), f1 = floorf(u1), f2 = floorf(u2);
                const float wx = u0 - f0, wy = u1 - f1, wz = u2 - f2;
                const int ix = (int)f0, iy = (int)f1, iz = (int)f2;
                const uint32_t x0 = (uint32_t)max(ix, 0), x1 = (uint32_t)min(ix + 1, mx);
                const uint32_t y0 = (uint32_t)max(iy, 0), y1 = (uint32_t)min(iy + 1, my);
                const uint32_t z0 = (uint32_t)max(iz, 0), z1 = (uint32_t)min(iz + 1, mz);
                const size_t r00 = ((size_t)z0 * Y + y0) * X, r10 = ((size_t)z0 * Y + y1) * X;
                const size_t r01 = ((size_t)z1 * Y + y0) * X, r11 = ((size_t)z1 * Y + y1) * X;
                const uint32_t v000 = g.vox[r00 + x0], v100 = g.vox[r00 + x1];
                const uint32_t v010 = g.vox[r10 + x0], v110 = g.vox[r10 + x1];
                const uint32_t v001 = g.vox[r01 + x0], v101 = g.vox[r01 + x1];
                const uint32_t v011 = g.vox[r11 + x0], v111 = g.vox[r11 + x1];

                auto tri = [&](int k) -> float {
                    const float a00 = ub(v000, k), a10 = ub(v010, k), a01 = ub(v001, k), a11 = ub(v011, k);
                    const float c00 = __builtin_fmaf(wx, ub(v100, k) - a00, a00);
                    const float c10 = __builtin_fmaf(wx, ub(v110, k) - a10, a10);
                    const float c01 = __builtin_fmaf(wx, ub(v101, k) - a01, a01);
                    const float c11 = __builtin_fmaf(wx, ub(v111, k) - a11, a11);
                    const float c0 = __builtin_fmaf(wy, c10 - c00, c00);
                    const float c1 = __builtin_fmaf(wy, c11 - c01, c01);
                    return __builtin_fmaf(wz, c1 - c0, c0);
                };
                const float a255 = tri(3);
                const float alpha_src = 1.0f - __builtin_amdgcn_exp2f(a255 * kexp);
                const float om = 1.0f - alpha;
                const float w = om * alpha_src;
                if (USE_TF) {
                    // wgsl:145-147: piecewise-linear in a with the breakpoint at 0.5
                    const float a = a255 * (1.0f / 255.0f);
                    const bool hi = a >= 0.5f;
                    const float t = hi ? (a - 0.5f) * 2.0f : a * 2.0f;
                    const float s0 = hi ? P.col_b[0] : P.col_a[0], e0 = hi ? P.col_c[0] : P.col_b[0];
                    const float s1 = hi ? P.col_b[1] : P.col_a[1], e1 = hi ? P.col_c[1] : P.col_b[1];
                    const float s2 = hi ? P.col_b[2] : P.col_a[2], e2 = hi ? P.col_c[2] : P.col_b[2];
                    cr = __builtin_fmaf(w, __builtin_fmaf(t, e0 - s0, s0), cr);
                    cg = __builtin_fmaf(w, __builtin_fmaf(t, e1 - s1, s1), cg);
                    cb = __builtin_fmaf(w, __builtin_fmaf(t, e2 - s2, s2), cb);
                } else if (ALPHA_ONLY) {
                    cr = __builtin_fmaf(w, a255, cr); // r == g == b == a for scalar volumes
                } else {
                    cr = __builtin_fmaf(w, tri(0), cr);
                    cg = __builtin_fmaf(w, tri(1), cg);
                    cb = __builtin_fmaf(w, tri(2), cb);
                }
                alpha = __builtin_fmaf(om, alpha_src, alpha);
                if (alpha >= 1.0f) break; // wgsl:176-178
            }
            dt += VRM_STEP_SIZE;          // wgsl:181, running f32 sum
            if (dt >= ray.t_max) break;   // wgsl:184-186
        }
        if (!USE_TF) {
            const float s = 1.0f / 255.0f;
            if (ALPHA_ONLY) { cr *= s; cg = cr; cb = cr; }
            else { cr *= s; cg *= s; cb *= s; }
        }
    }

    const uint32_t packed = unorm8(cr) | (unorm8(cg) << 8) | (unorm8(cb) << 16) | (unorm8(alpha) << 24);
    v.out[(size_t)py * P.pitch_px + px] = packed;
    if (DIAG) {
        const size_t o = (size_t)py * P.width + px;
        if (v.diag_rgba) {
            float4 c = make_float4(cr, cg, cb, alpha);
            reinterpret_cast<float4 *>(v.diag_rgba)[o] = c;
        }
        if (v.diag_iters) v.diag_iters[o] = iters;
        if (v.diag_geom) {
            float4 *gq = reinterpret_cast<float4 *>(v.diag_geom) + 2 * o;
            gq[0] = make_float4(ray.rd[0], ray.rd[1], ray.rd[2], ray.hit ? ray.t_min : 0.0f);
            gq[1] = make_float4(ray.hit ? ray.t_max : 0.0f, (float)n_inside, ray.hit ? 1.0f : 0.0f, (float)iters);
        }
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

// loaders/dat.rs:62-74: big-endian combine, ((val / (2^(8n)-1)) * 255) as u8, replicated to RGBA
__global__ void __launch_bounds__(256) quantize_scalar_kernel(const uint8_t *__restrict__ raw, uint32_t bps, size_t n,
                                                              uint32_t *__restrict__ dst)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float val, denom;
        if (bps == 2) {
            val = (float)(((uint32_t)raw[2 * i] << 8) | raw[2 * i + 1]);
            denom = 65535.0f;
        } else {
            val = (float)raw[i];
            denom = 255.0f;
        }
        float q = ((val) / (denom)) * 255.0f;
        q = fminf(fmaxf(q, 0.0f), 255.0f);
        dst[i] = (uint32_t)q * 0x01010101u;
    }
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

inline uint32_t stream_blocks(size_t n)
{
    size_t b = (n + 255) / 256;
    const size_t cap = 256 * 16; // 256 CUs x 16 blocks keeps every CU streaming
    return (uint32_t)(b < cap ? (b ? b : 1) : cap);
}

} // namespace

hipError_t vrm_launch_march(const VrmMarchParams &p, uint32_t n_views, int variant, bool diagnostics, hipStream_t stream)
{
    (void)variant;
    const dim3 grid(p.tiles_x * p.tiles_y, n_views, 1), block(256, 1, 1);
    const bool tf = p.use_tf != 0;
    const bool alpha_only = p.grid.scalar != 0;
#define VRM_LAUNCH(A, T, D) hipLaunchKernelGGL((march_linear_kernel<A, T, D>), grid, block, 0, stream, p)
    if (diagnostics) {
        if (tf) VRM_LAUNCH(true, true, true);
        else if (alpha_only) VRM_LAUNCH(true, false, true);
        else VRM_LAUNCH(false, false, true);
    } else {
        if (tf) VRM_LAUNCH(true, true, false);
        else if (alpha_only) VRM_LAUNCH(true, false, false);
        else VRM_LAUNCH(false, false, false);
    }
#undef VRM_LAUNCH
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
