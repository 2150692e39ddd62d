/*
 * vr_oracle.c -- CPU restatement of the ray-marching hot path of bennetde/volume-renderer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as
 * the checker (or as the timed CPU baseline), never as a fallback for the HIP path.
 *
 * PARITY UNPINNED: the reference ships no tests, golden images or fixtures for this path
 *  and cannot be built or run here (no Rust toolchain, no Vulkan ICD).
 * The arithmetic of three third-party pieces is restated from their published algorithms:
 *   - glam 0.28.0 (Cargo.lock:1136)  Mat4::look_to_rh / perspective_rh / inverse,
 *     Quat::from_mat4 / conjugate / mul_vec3           (call sites camera.rs:26-31,66; transform.rs:30-32,59-63)
 *   - the GPU sampler behind wgpu 0.19 textureSample   (call site raymarcher.wgsl:227;
 *     sampler state texture_3d.rs:66-76 => linear filter, clamp-to-edge, texel centres at +0.5)
 *   - Rgba8Unorm decode / encode                        (texture_3d.rs:43; ray_marcher/mod.rs:79-83)
 * The pins are the analytic known-answer tests in tests/test_oracle_known_answers.py and an
 * independently written numpy twin (oracle/oracle_np.py).
 *
 * Floating point: every operation is a separately rounded IEEE binary32 operation, in the order
 * the WGSL source writes it (build with -ffp-contract=off, no fast-math).  The HIP kernels are
 * required to reproduce the *geometric* quantities (ray direction, t_min/t_max, the running dt
 * sum, the sample position and the in-box test) bit for bit, because those feed discontinuous
 * decisions; filter / transfer function / blend arithmetic only has to agree to rounding.
 *
 * Reference files followed (all under /root/reference/src):
 *   ray_marcher/raymarcher.wgsl:98-299   fs_main, raymarch, scene, remap, aabb_intersect
 *   ray_marcher/mod.rs:8-16,79-83,124    fullscreen triangle uv, REPLACE blend, transparent clear
 *   voxel/grid.rs:265-298                VoxelGridUniform::new, RaymarchTransferFunctionColors::new
 *   voxel/voxel.rs:13-16                 default voxel [0,0,0,255]
 *   camera.rs:15-67, transform.rs:30-63  camera matrices
 *   camera_sphere_controller.rs:73-93    sphere positions
 *   loaders/dat.rs:58-76, loaders/netcdf.rs:49-61   quantisation to RGBA8
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* small f32 vector helpers (each op separately rounded)                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;
typedef struct { float c[4][4]; } m4; /* c[col][row], column-major like glam::Mat4 */

static v3 v3_make(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 v3_scale(v3 a, float s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static v3 v3_cross(v3 a, v3 b)
{
    return v3_make(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
/* glam Vec3::normalize = self * (1 / sqrt(dot(self, self))) */
static v3 v3_normalize(v3 a) { return v3_scale(a, 1.0f / sqrtf(v3_dot(a, a))); }

/* ------------------------------------------------------------------------------------------ */
/* glam restatements (camera.rs:26-31, transform.rs:30-32,59-63)                                */
/* ------------------------------------------------------------------------------------------ */

/* Mat4::look_to_rh(eye, dir, up): f = normalize(dir), s = normalize(f x up), u = s x f */
static m4 look_to_rh(v3 eye, v3 dir, v3 up)
{
    v3 f = v3_normalize(dir);
    v3 s = v3_normalize(v3_cross(f, up));
    v3 u = v3_cross(s, f);
    m4 m;
    m.c[0][0] = s.x; m.c[0][1] = u.x; m.c[0][2] = -f.x; m.c[0][3] = 0.0f;
    m.c[1][0] = s.y; m.c[1][1] = u.y; m.c[1][2] = -f.y; m.c[1][3] = 0.0f;
    m.c[2][0] = s.z; m.c[2][1] = u.z; m.c[2][2] = -f.z; m.c[2][3] = 0.0f;
    m.c[3][0] = -v3_dot(eye, s); m.c[3][1] = -v3_dot(eye, u); m.c[3][2] = v3_dot(eye, f); m.c[3][3] = 1.0f;
    return m;
}

/* Mat4::perspective_rh(fovy, aspect, near, far): right-handed, depth range [0,1] */
static m4 perspective_rh(float fovy, float aspect, float z_near, float z_far)
{
    float half = 0.5f * fovy;
    float sn = sinf(half), cs = cosf(half);
    float h = cs / sn;
    float w = h / aspect;
    float r = z_far / (z_near - z_far);
    m4 m;
    memset(&m, 0, sizeof m);
    m.c[0][0] = w;
    m.c[1][1] = h;
    m.c[2][2] = r; m.c[2][3] = -1.0f;
    m.c[3][2] = r * z_near;
    return m;
}

static m4 m4_mul(m4 a, m4 b)
{
    /* column j of a*b = a.col0*b[j][0] + a.col1*b[j][1] + a.col2*b[j][2] + a.col3*b[j][3] */
    m4 r;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.c[j][i] = ((a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1]) + a.c[2][i] * b.c[j][2]) + a.c[3][i] * b.c[j][3];
    return r;
}

/* General 4x4 inverse by 2x2 sub-determinants (the GLM/glam cofactor scheme), times 1/det. */
static m4 m4_inverse(m4 m)
{
    const float (*a)[4] = m.c; /* a[col][row] */
    float s00 = a[2][2] * a[3][3] - a[3][2] * a[2][3];
    float s02 = a[1][2] * a[3][3] - a[3][2] * a[1][3];
    float s03 = a[1][2] * a[2][3] - a[2][2] * a[1][3];
    float s04 = a[2][1] * a[3][3] - a[3][1] * a[2][3];
    float s06 = a[1][1] * a[3][3] - a[3][1] * a[1][3];
    float s07 = a[1][1] * a[2][3] - a[2][1] * a[1][3];
    float s08 = a[2][1] * a[3][2] - a[3][1] * a[2][2];
    float s10 = a[1][1] * a[3][2] - a[3][1] * a[1][2];
    float s11 = a[1][1] * a[2][2] - a[2][1] * a[1][2];
    float s12 = a[2][0] * a[3][3] - a[3][0] * a[2][3];
    float s14 = a[1][0] * a[3][3] - a[3][0] * a[1][3];
    float s15 = a[1][0] * a[2][3] - a[2][0] * a[1][3];
    float s16 = a[2][0] * a[3][2] - a[3][0] * a[2][2];
    float s18 = a[1][0] * a[3][2] - a[3][0] * a[1][2];
    float s19 = a[1][0] * a[2][2] - a[2][0] * a[1][2];
    float s20 = a[2][0] * a[3][1] - a[3][0] * a[2][1];
    float s22 = a[1][0] * a[3][1] - a[3][0] * a[1][1];
    float s23 = a[1][0] * a[2][1] - a[2][0] * a[1][1];

    float f0[4] = { s00, s00, s02, s03 }, f1[4] = { s04, s04, s06, s07 }, f2[4] = { s08, s08, s10, s11 };
    float f3[4] = { s12, s12, s14, s15 }, f4[4] = { s16, s16, s18, s19 }, f5[4] = { s20, s20, s22, s23 };
    float v0[4] = { a[1][0], a[0][0], a[0][0], a[0][0] };
    float v1[4] = { a[1][1], a[0][1], a[0][1], a[0][1] };
    float v2[4] = { a[1][2], a[0][2], a[0][2], a[0][2] };
    float v3_[4] = { a[1][3], a[0][3], a[0][3], a[0][3] };
    static const float sa[4] = { 1.0f, -1.0f, 1.0f, -1.0f }, sb[4] = { -1.0f, 1.0f, -1.0f, 1.0f };

    m4 inv;
    for (int i = 0; i < 4; ++i) {
        inv.c[0][i] = ((v1[i] * f0[i] - v2[i] * f1[i]) + v3_[i] * f2[i]) * sa[i];
        inv.c[1][i] = ((v0[i] * f0[i] - v2[i] * f3[i]) + v3_[i] * f4[i]) * sb[i];
        inv.c[2][i] = ((v0[i] * f1[i] - v1[i] * f3[i]) + v3_[i] * f5[i]) * sa[i];
        inv.c[3][i] = ((v0[i] * f2[i] - v1[i] * f4[i]) + v2[i] * f5[i]) * sb[i];
    }
    float d0 = a[0][0] * inv.c[0][0], d1 = a[0][1] * inv.c[1][0], d2 = a[0][2] * inv.c[2][0], d3 = a[0][3] * inv.c[3][0];
    float det = ((d0 + d1) + d2) + d3;
    float rcp = 1.0f / det;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i) inv.c[j][i] *= rcp;
    return inv;
}

typedef struct { float x, y, z, w; } quat;

/* Quat::from_mat4: rotation matrix (upper 3x3, columns = axes) -> quaternion, the branch
 * scheme of M. Day, "Converting a Rotation Matrix to a Quaternion" (used by glam). */
static quat quat_from_mat4(m4 m)
{
    float m00 = m.c[0][0], m01 = m.c[0][1], m02 = m.c[0][2];
    float m10 = m.c[1][0], m11 = m.c[1][1], m12 = m.c[1][2];
    float m20 = m.c[2][0], m21 = m.c[2][1], m22 = m.c[2][2];
    quat q;
    if (m22 <= 0.0f) {
        float dif10 = m11 - m00, omm22 = 1.0f - m22;
        if (dif10 <= 0.0f) {
            float four_xsq = omm22 - dif10, k = 0.5f / sqrtf(four_xsq);
            q.x = four_xsq * k; q.y = (m01 + m10) * k; q.z = (m02 + m20) * k; q.w = (m12 - m21) * k;
        } else {
            float four_ysq = omm22 + dif10, k = 0.5f / sqrtf(four_ysq);
            q.x = (m01 + m10) * k; q.y = four_ysq * k; q.z = (m12 + m21) * k; q.w = (m20 - m02) * k;
        }
    } else {
        float sum10 = m11 + m00, opm22 = 1.0f + m22;
        if (sum10 <= 0.0f) {
            float four_zsq = opm22 - sum10, k = 0.5f / sqrtf(four_zsq);
            q.x = (m02 + m20) * k; q.y = (m12 + m21) * k; q.z = four_zsq * k; q.w = (m01 - m10) * k;
        } else {
            float four_wsq = opm22 + sum10, k = 0.5f / sqrtf(four_wsq);
            q.x = (m12 - m21) * k; q.y = (m20 - m02) * k; q.z = (m01 - m10) * k; q.w = four_wsq * k;
        }
    }
    return q;
}

/* Quat * Vec3 (glam): v*(w^2 - b.b) + b*(2 v.b) + (b x v)*(2w),  b = (x,y,z) */
static v3 quat_rotate(quat q, v3 v)
{
    v3 b = v3_make(q.x, q.y, q.z);
    float b2 = v3_dot(b, b);
    v3 t0 = v3_scale(v, q.w * q.w - b2);
    v3 t1 = v3_scale(b, v3_dot(v, b) * 2.0f);
    v3 t2 = v3_scale(v3_cross(b, v), q.w * 2.0f);
    return v3_add(v3_add(t0, t1), t2);
}

/* Transform::look_to (transform.rs:59-63): rotation = conj(from_mat4(look_to_rh(pos, normalize(target-pos), up))) */
static quat transform_look_to(v3 position, v3 target, v3 up)
{
    v3 d = v3_normalize(v3_sub(target, position));
    quat q = quat_from_mat4(look_to_rh(position, d, up));
    q.x = -q.x; q.y = -q.y; q.z = -q.z;
    return q;
}

/*
 * CameraUniform::update_view_proj (camera.rs:61-67) for a camera at `position` that was aimed with
 * Transform::look_to(target, +Y) (camera_sphere_controller.rs:73-76, state.rs:231).
 * out36 = position[4] (w = 0), view_proj[16], inverse_view_proj[16], column-major.
 * basis9 (optional) = right, up, forward as written to cameras.json (sphere_screenshot_manager.rs:56-64).
 */
ORC_API void orc_camera_uniform(const float position[3], const float target[3], float aspect, float out36[36],
                                float basis9[9])
{
    v3 pos = v3_make(position[0], position[1], position[2]);
    v3 tgt = v3_make(target[0], target[1], target[2]);
    quat rot = transform_look_to(pos, tgt, v3_make(0.0f, 1.0f, 0.0f));
    v3 forward = quat_rotate(rot, v3_make(0.0f, 0.0f, -1.0f));           /* transform.rs:30-32 */
    m4 view = look_to_rh(pos, forward, v3_make(0.0f, 1.0f, 0.0f));       /* camera.rs:27 */
    const float fovy_deg = 45.0f;                                        /* camera.rs:19 */
    float fovy_rad = fovy_deg * (3.14159265358979323846f / 180.0f);      /* f32::to_radians */
    m4 proj = perspective_rh(fovy_rad, aspect, 0.1f, 1000.0f);           /* camera.rs:20-21,28 */
    m4 vp = m4_mul(proj, view);                                          /* camera.rs:30 */
    m4 ivp = m4_inverse(vp);                                             /* camera.rs:66 */
    out36[0] = pos.x; out36[1] = pos.y; out36[2] = pos.z; out36[3] = 0.0f;
    memcpy(out36 + 4, vp.c, 64);
    memcpy(out36 + 20, ivp.c, 64);
    if (basis9) {
        v3 r = quat_rotate(rot, v3_make(1.0f, 0.0f, 0.0f));
        v3 u = quat_rotate(rot, v3_make(0.0f, 1.0f, 0.0f));
        basis9[0] = r.x; basis9[1] = r.y; basis9[2] = r.z;
        basis9[3] = u.x; basis9[4] = u.y; basis9[5] = u.z;
        basis9[6] = forward.x; basis9[7] = forward.y; basis9[8] = forward.z;
    }
}

/* CameraSphereController::get_position_on_sphere (camera_sphere_controller.rs:79-89) */
ORC_API void orc_sphere_position(uint32_t ix, uint32_t iy, uint32_t xdiv, uint32_t ydiv, float radius,
                                 const float origin[3], float out[3])
{
    const float PI = 3.14159265358979323846f;
    float theta = (float)ix / (float)xdiv * PI * 2.0f;
    float phi = (float)iy / (float)ydiv * PI;
    float x = sinf(phi) * cosf(theta) * radius;
    float y = sinf(phi) * sinf(theta) * radius;
    float z = cosf(phi) * radius;
    out[0] = x + origin[0];
    out[1] = z + origin[1];
    out[2] = -y + origin[2];
}

/* VoxelGridUniform::new (grid.rs:265-280). out16 words: dims[4] (u32 bit patterns), box_min[4], box_size[4], buffer[4] */
ORC_API void orc_grid_uniform(const uint32_t dims[3], float attenuation, uint32_t out_dims[4], float out_box_min[4],
                              float out_box_size[4], float out_buffer[4])
{
    uint32_t mn = dims[0] < dims[1] ? dims[0] : dims[1];
    if (dims[2] < mn) mn = dims[2];
    for (int i = 0; i < 3; ++i) {
        out_box_size[i] = (float)dims[i] / (float)mn;
        out_box_min[i] = -out_box_size[i] / 2.0f;
        out_dims[i] = dims[i];
    }
    out_dims[3] = dims[0];
    out_box_min[3] = 0.0f;
    out_box_size[3] = 0.0f;
    for (int i = 0; i < 4; ++i) out_buffer[i] = attenuation;
}

/* ------------------------------------------------------------------------------------------ */
/* quantisation to RGBA8 (loaders)                                                              */
/* ------------------------------------------------------------------------------------------ */

/* Rust `f32 as u8`: truncate toward zero, saturate, NaN -> 0 */
static uint8_t rust_f32_as_u8(float v)
{
    if (!(v == v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}

/* loaders/dat.rs:58-76. raw = UCHAR (bytes_per_sample 1) or USHORT combined big-endian (2). out = [b,b,b,b] per voxel */
ORC_API void orc_quantize_dat(const uint8_t *raw, size_t n_voxels, int bytes_per_sample, uint8_t *out_rgba8)
{
    float denom = (bytes_per_sample == 2) ? 65535.0f : 255.0f;
    for (size_t i = 0; i < n_voxels; ++i) {
        size_t val = 0;
        for (int b = 0; b < bytes_per_sample; ++b) {
            val |= raw[i * (size_t)bytes_per_sample + (size_t)b];
            if (b < bytes_per_sample - 1) val <<= 8;
        }
        uint8_t q = rust_f32_as_u8(((float)val / denom) * 255.0f);
        out_rgba8[4 * i + 0] = q; out_rgba8[4 * i + 1] = q; out_rgba8[4 * i + 2] = q; out_rgba8[4 * i + 3] = q;
    }
}

/* loaders/netcdf.rs:49-61. data = f32 color(c=4, z, y, x); out voxel (x,y,z) at x + X*(y + Y*z) */
ORC_API void orc_quantize_netcdf(const float *data, uint32_t X, uint32_t Y, uint32_t Z, uint8_t *out_rgba8)
{
    size_t n = (size_t)X * Y * Z;
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 4; ++c) out_rgba8[4 * i + (size_t)c] = rust_f32_as_u8(data[(size_t)c * n + i] * 255.0f);
}

/* ------------------------------------------------------------------------------------------ */
/* the shader                                                                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint8_t *voxels; /* RGBA8, idx = x + X*(y + Y*z) (grid.rs:211) */
    uint32_t dim[3];
    float box_min[3], box_size[3];
    float attenuation;     /* voxel_grid.buffer[0] (wgsl:150) */
    float col_a[3], col_b[3], col_c[3];
    int use_tf;
    const float *cam;      /* 36 floats */
    /* sort-last partial renders (no reference counterpart): with slab != 0 only samples whose position
     * has z in [slab_lo, slab_hi) contribute; every other sample reads as scene() == 0, i.e. adds exactly nothing */
    int slab;
    float slab_lo, slab_hi;
    /* test-infrastructure form of the same slab render for volumes that do not fit host memory whole: `voxels` holds only
     * the z slices [z_off, z_off + z_cnt) -- what a slab's samples can touch -- and, with `scalar`, one byte per voxel
     * standing for [b,b,b,b] (dat.rs:74).  A texel outside the stored range reads as NaN so that a mistake shows. */
    uint32_t z_off, z_cnt;
    int scalar;
} scene_t;

#define MAX_STEP_AMOUNT 5000 /* wgsl:96 */
static const float STEP_SIZE = 0.001f; /* wgsl:126 */

static float texel(const scene_t *s, int x, int y, int z, int ch)
{
    if ((uint32_t)z < s->z_off || (uint32_t)z - s->z_off >= s->z_cnt) return NAN;
    size_t idx = (size_t)x + (size_t)s->dim[0] * ((size_t)y + (size_t)s->dim[1] * (size_t)((uint32_t)z - s->z_off));
    if (s->scalar) return (float)s->voxels[idx] / 255.0f;
    return (float)s->voxels[4 * idx + (size_t)ch] / 255.0f; /* Rgba8Unorm decode */
}

static float lerpf(float a, float b, float w) { return a + w * (b - a); }

/* textureSample with linear/linear filter, clamp-to-edge (texture_3d.rs:66-76): texel centres at
 * (i + 0.5)/dim, so the continuous texel coordinate of normalised coordinate r is r*dim - 0.5. */
static void texture_sample(const scene_t *s, const float rel[3], float out[4])
{
    int i0[3], i1[3];
    float fr[3];
    for (int a = 0; a < 3; ++a) {
        float u = rel[a] * (float)s->dim[a] - 0.5f;
        float fl = floorf(u);
        fr[a] = u - fl;
        int lo = (int)fl, hi = lo + 1, mx = (int)s->dim[a] - 1;
        i0[a] = lo < 0 ? 0 : (lo > mx ? mx : lo);
        i1[a] = hi < 0 ? 0 : (hi > mx ? mx : hi);
    }
    for (int ch = 0; ch < 4; ++ch) {
        float c00 = lerpf(texel(s, i0[0], i0[1], i0[2], ch), texel(s, i1[0], i0[1], i0[2], ch), fr[0]);
        float c10 = lerpf(texel(s, i0[0], i1[1], i0[2], ch), texel(s, i1[0], i1[1], i0[2], ch), fr[0]);
        float c01 = lerpf(texel(s, i0[0], i0[1], i1[2], ch), texel(s, i1[0], i0[1], i1[2], ch), fr[0]);
        float c11 = lerpf(texel(s, i0[0], i1[1], i1[2], ch), texel(s, i1[0], i1[1], i1[2], ch), fr[0]);
        float c0 = lerpf(c00, c10, fr[1]);
        float c1 = lerpf(c01, c11, fr[1]);
        out[ch] = lerpf(c0, c1, fr[2]);
    }
}

/* scene() wgsl:199-238. returns rgb + alpha; zeros outside the half-open box */
static void scene(const scene_t *s, const float p[3], float out[4])
{
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    float box_max[3], rel[3];
    for (int a = 0; a < 3; ++a) {
        box_max[a] = s->box_min[a] + s->box_size[a];
        rel[a] = (p[a] - s->box_min[a]) / s->box_size[a];
    }
    if (p[0] < s->box_min[0] || p[1] < s->box_min[1] || p[2] < s->box_min[2]) return;
    if (p[0] >= box_max[0] || p[1] >= box_max[1] || p[2] >= box_max[2]) return;
    if (s->slab && (p[2] < s->slab_lo || p[2] >= s->slab_hi)) return;
    texture_sample(s, rel, out);
}

/* remap() wgsl:240-242 */
static float remap(float v, float min1, float max1, float min2, float max2)
{
    return min2 + (v - min1) * (max2 - min2) / (max1 - min1);
}
/* WGSL mix(e1,e2,e3) = e1*(1-e3) + e2*e3 */
static float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

/* aabb_intersect() wgsl:247-299 */
static int aabb_intersect(const float ro[3], const float rd[3], const float bmin[3], const float bsize[3], float *tmin_out,
                          float *tmax_out)
{
    float mx[3] = { bmin[0] + bsize[0], bmin[1] + bsize[1], bmin[2] + bsize[2] };
    float t_min = (bmin[0] - ro[0]) / rd[0];
    float t_max = (mx[0] - ro[0]) / rd[0];
    if (t_min > t_max) { float t = t_min; t_min = t_max; t_max = t; }
    float t_y_min = (bmin[1] - ro[1]) / rd[1];
    float t_y_max = (mx[1] - ro[1]) / rd[1];
    if (t_y_min > t_y_max) { float t = t_y_min; t_y_min = t_y_max; t_y_max = t; }
    if ((t_min > t_y_max) || (t_y_min > t_max)) return 0;
    if (t_y_min > t_min) t_min = t_y_min;
    if (t_y_max < t_max) t_max = t_y_max;
    float t_z_min = (bmin[2] - ro[2]) / rd[2];
    float t_z_max = (mx[2] - ro[2]) / rd[2];
    if (t_z_min > t_z_max) { float t = t_z_min; t_z_min = t_z_max; t_z_max = t; }
    if ((t_min > t_z_max) || (t_z_min > t_max)) return 0;
    if (t_z_min > t_min) t_min = t_z_min;
    if (t_z_max < t_max) t_max = t_z_max;
    *tmin_out = t_min;
    *tmax_out = t_max;
    return 1;
}

/* raymarch() wgsl:119-196. returns number of loop iterations executed (0 on a miss) */
static uint32_t raymarch(const scene_t *s, const float ro[3], const float rd[3], float out[4])
{
    float color[3] = { 0.0f, 0.0f, 0.0f }, alpha = 0.0f;
    float t_min, t_max;
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    if (!aabb_intersect(ro, rd, s->box_min, s->box_size, &t_min, &t_max)) return 0; /* wgsl:131-134 */
    float dt = t_min; /* wgsl:137: not clamped, may be negative */
    uint32_t iters = 0;
    for (int i = 0; i < MAX_STEP_AMOUNT; ++i) {
        ++iters;
        float p[3] = { ro[0] + rd[0] * dt, ro[1] + rd[1] * dt, ro[2] + rd[2] * dt };
        float hit[4];
        scene(s, p, hit);
        float a = hit[3];
        float alpha_src = 1.0f - expf(-a * STEP_SIZE * s->attenuation); /* wgsl:150 */
        float src[3];
        if (s->use_tf) { /* wgsl:145-147,153-154 */
            float t_ab = remap(a, 0.0f, 0.5f, 0.0f, 1.0f);
            float t_bc = remap(a, 0.5f, 1.0f, 0.0f, 1.0f);
            float st = (a >= 0.5f) ? 1.0f : 0.0f; /* step(0.5, a) */
            for (int c = 0; c < 3; ++c) {
                float ab = mixf(s->col_a[c], s->col_b[c], t_ab);
                float bc = mixf(s->col_b[c], s->col_c[c], t_bc);
                src[c] = mixf(ab, bc, st);
            }
        } else {
            src[0] = hit[0]; src[1] = hit[1]; src[2] = hit[2];
        }
        float w = (1.0f - alpha) * alpha_src;
        for (int c = 0; c < 3; ++c) color[c] = color[c] + w * src[c]; /* wgsl:158 */
        alpha = alpha + (1.0f - alpha) * alpha_src;                    /* wgsl:159 */
        if (alpha >= 1.0f) break;                                      /* wgsl:176-178 */
        dt += STEP_SIZE;                                               /* wgsl:181 */
        if (dt >= t_max) break;                                        /* wgsl:184-186 */
    }
    out[0] = color[0]; out[1] = color[1]; out[2] = color[2]; out[3] = alpha;
    return iters;
}

/* Rgba8Unorm store: round-to-nearest of clamp(v,0,1)*255 (mod.rs:79-83, state.rs:78) */
static uint8_t unorm8(float v)
{
    if (!(v == v)) return 0;
    if (v < 0.0f) v = 0.0f;
    if (v > 1.0f) v = 1.0f;
    return (uint8_t)floorf(v * 255.0f + 0.5f);
}

/* fs_main() wgsl:98-116 for framebuffer pixel (px,py), origin top-left.  The fullscreen triangle
 * (mod.rs:8-12) interpolates tex_coords to uv = (ndc + 1)/2 at the pixel centre. */
static uint32_t shade_pixel(const scene_t *s, uint32_t px, uint32_t py, uint32_t W, uint32_t H, float out[4])
{
    float uvx = ((float)px + 0.5f) / (float)W;
    float uvy = 1.0f - ((float)py + 0.5f) / (float)H;
    float sx = uvx * 2.0f - 1.0f, sy = uvy * 2.0f - 1.0f, sz = 1.0f, sw = 1.0f; /* wgsl:101-103 */
    const float *iv = s->cam + 20; /* inverse_view_proj, column-major */
    float w3[3];
    for (int r = 0; r < 3; ++r) /* wgsl:106: M*v = c0*x + c1*y + c2*z + c3*w */
        w3[r] = ((iv[0 + r] * sx + iv[4 + r] * sy) + iv[8 + r] * sz) + iv[12 + r] * sw;
    float len = sqrtf((w3[0] * w3[0] + w3[1] * w3[1]) + w3[2] * w3[2]);
    float rd[3] = { w3[0] / len, w3[1] / len, w3[2] / len }; /* normalize, wgsl:107 */
    float ro[3] = { s->cam[0], s->cam[1], s->cam[2] };       /* wgsl:108 */
    return raymarch(s, ro, rd, out);
}

/*
 * Render one view.  voxels: RGBA8 x-fastest.  cam36: CameraUniform.  tf16: RaymarchTransferFunctionColors
 * as 16 32-bit words (12 floats + use_transfer_function[4] as u32 bit patterns).
 * out_f32 (W*H*4), out_rgba8 (W*H*4), out_steps (W*H) are each optional (NULL to skip).
 * Returns the total number of loop iterations over all rays (the reference-equivalent sample count).
 */
typedef struct {
    uint32_t z_off, z_cnt; /* stored slices (the whole volume: 0, dims[2]) */
    int scalar;            /* 1 byte per voxel */
    uint32_t crop[4];      /* x0, y0, w, h of the pixels to render; the outputs are w*h */
} render_opts;

static uint64_t render_impl(const uint8_t *voxels, const uint32_t dims[3], const float cam36[36], const uint32_t tf16[16],
                            float attenuation, uint32_t W, uint32_t H, float *out_f32, uint8_t *out_rgba8,
                            uint32_t *out_steps, int n_threads, int slab, uint32_t z_begin, uint32_t z_end, const render_opts *ro)
{
    scene_t s;
    memset(&s, 0, sizeof s);
    uint32_t d4[4];
    float bmin[4], bsz[4], buf[4];
    orc_grid_uniform(dims, attenuation, d4, bmin, bsz, buf);
    s.voxels = voxels;
    for (int i = 0; i < 3; ++i) { s.dim[i] = dims[i]; s.box_min[i] = bmin[i]; s.box_size[i] = bsz[i]; }
    s.attenuation = buf[0];
    const float *tf = (const float *)tf16;
    for (int i = 0; i < 3; ++i) { s.col_a[i] = tf[i]; s.col_b[i] = tf[4 + i]; s.col_c[i] = tf[8 + i]; }
    s.use_tf = tf16[12] != 0;
    s.cam = cam36;
    s.z_off = ro ? ro->z_off : 0;
    s.z_cnt = ro ? ro->z_cnt : dims[2];
    s.scalar = ro ? ro->scalar : 0;
    const uint32_t cx0 = ro ? ro->crop[0] : 0, cy0 = ro ? ro->crop[1] : 0, cw = ro ? ro->crop[2] : W, chh = ro ? ro->crop[3] : H;
    if (slab) {
        /* world-space bounds of the slices [z_begin, z_end): the box faces at the ends, one shared f32 value between */
        const float Zf = (float)dims[2];
        s.slab = 1;
        s.slab_lo = z_begin == 0 ? s.box_min[2] : s.box_min[2] + ((float)z_begin / Zf) * s.box_size[2];
        s.slab_hi = z_end == dims[2] ? s.box_min[2] + s.box_size[2] : s.box_min[2] + ((float)z_end / Zf) * s.box_size[2];
    }

    uint64_t total = 0;
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads) reduction(+ : total)
    for (int64_t qy = 0; qy < (int64_t)chh; ++qy) {
        for (uint32_t qx = 0; qx < cw; ++qx) {
            float c[4];
            const uint32_t px = cx0 + qx, py = cy0 + (uint32_t)qy;
            uint32_t it = shade_pixel(&s, px, py, W, H, c);
            size_t o = (size_t)qy * cw + qx;
            total += it;
            if (out_steps) out_steps[o] = it;
            if (out_f32) { out_f32[4 * o] = c[0]; out_f32[4 * o + 1] = c[1]; out_f32[4 * o + 2] = c[2]; out_f32[4 * o + 3] = c[3]; }
            if (out_rgba8) for (int k = 0; k < 4; ++k) out_rgba8[4 * o + (size_t)k] = unorm8(c[k]);
        }
    }
    return total;
}

ORC_API uint64_t orc_render(const uint8_t *voxels, const uint32_t dims[3], const float cam36[36], const uint32_t tf16[16],
                            float attenuation, uint32_t W, uint32_t H, float *out_f32, uint8_t *out_rgba8,
                            uint32_t *out_steps, int n_threads)
{
    return render_impl(voxels, dims, cam36, tf16, attenuation, W, H, out_f32, out_rgba8, out_steps, n_threads, 0, 0, 0, NULL);
}

/* The partial image of the z slab [z_begin, z_end) of a sort-last render: the whole volume's sample lattice, only the
 * slab's samples blended.  Compositing the partials of a partition with "over" in ray order gives orc_render's image
 * up to f32 re-association. */
ORC_API uint64_t orc_render_slab(const uint8_t *voxels, const uint32_t dims[3], const float cam36[36], const uint32_t tf16[16],
                                 float attenuation, uint32_t W, uint32_t H, uint32_t z_begin, uint32_t z_end, float *out_f32,
                                 int n_threads)
{
    return render_impl(voxels, dims, cam36, tf16, attenuation, W, H, out_f32, NULL, NULL, n_threads, 1, z_begin, z_end, NULL);
}

/* The same partial image from the STORED slices alone -- `stored` holds the slices [z_store0, z_store1) of a dims volume,
 * one byte per voxel if `scalar` (a DAT density, [b,b,b,b]) else RGBA8 -- and only for the pixels of crop = {x0, y0, w, h}
 * of the W x H image (outputs are w*h).  For volumes whose whole does not fit host memory (C5: 2048 x 2048 x 4096): a rank's
 * slab plus its two ghost slices is everything its samples can touch.  out_steps: loop iterations per ray (whole lattice). */
ORC_API uint64_t orc_render_slab_stored(const uint8_t *stored, int scalar, const uint32_t dims[3], uint32_t z_store0,
                                        uint32_t z_store1, const float cam36[36], const uint32_t tf16[16], float attenuation,
                                        uint32_t W, uint32_t H, uint32_t z_begin, uint32_t z_end, const uint32_t crop[4],
                                        float *out_f32, uint32_t *out_steps, int n_threads)
{
    render_opts ro;
    ro.z_off = z_store0;
    ro.z_cnt = z_store1 - z_store0;
    ro.scalar = scalar;
    for (int i = 0; i < 4; ++i) ro.crop[i] = crop[i];
    const int whole = z_begin == 0 && z_end == dims[2];
    return render_impl(stored, dims, cam36, tf16, attenuation, W, H, out_f32, NULL, out_steps, n_threads, whole ? 0 : 1, z_begin,
                       z_end, &ro);
}

/* Point sample for unit tests: scene(p) with the given grid (wgsl:199-238) */
ORC_API void orc_scene(const uint8_t *voxels, const uint32_t dims[3], const float p[3], float out[4])
{
    scene_t s;
    uint32_t d4[4];
    float bmin[4], bsz[4], buf[4];
    orc_grid_uniform(dims, 1.0f, d4, bmin, bsz, buf);
    memset(&s, 0, sizeof s);
    s.voxels = voxels;
    s.z_cnt = dims[2];
    for (int i = 0; i < 3; ++i) { s.dim[i] = dims[i]; s.box_min[i] = bmin[i]; s.box_size[i] = bsz[i]; }
    scene(&s, p, out);
}

/* Ray geometry for unit tests: direction, hit flag, t_min, t_max of pixel (px,py) */
ORC_API int orc_ray(const float cam36[36], const uint32_t dims[3], uint32_t px, uint32_t py, uint32_t W, uint32_t H,
                    float rd_out[3], float *t_min, float *t_max)
{
    uint32_t d4[4];
    float bmin[4], bsz[4], buf[4];
    orc_grid_uniform(dims, 1.0f, d4, bmin, bsz, buf);
    float uvx = ((float)px + 0.5f) / (float)W;
    float uvy = 1.0f - ((float)py + 0.5f) / (float)H;
    float sx = uvx * 2.0f - 1.0f, sy = uvy * 2.0f - 1.0f;
    const float *iv = cam36 + 20;
    float w3[3];
    for (int r = 0; r < 3; ++r) w3[r] = ((iv[0 + r] * sx + iv[4 + r] * sy) + iv[8 + r] * 1.0f) + iv[12 + r] * 1.0f;
    float len = sqrtf((w3[0] * w3[0] + w3[1] * w3[1]) + w3[2] * w3[2]);
    float rd[3] = { w3[0] / len, w3[1] / len, w3[2] / len };
    float ro[3] = { cam36[0], cam36[1], cam36[2] };
    rd_out[0] = rd[0]; rd_out[1] = rd[1]; rd_out[2] = rd[2];
    *t_min = 0.0f; *t_max = 0.0f;
    return aabb_intersect(ro, rd, bmin, bsz, t_min, t_max);
}

/*
 * Geometry of every ray of a view, 8 floats per pixel: rd.xyz, t_min, t_max, number of loop iterations
 * whose sample position passes the in-box test of scene() (wgsl:218-223) when the loop runs to t_max
 * (no early termination), hit flag, total iterations.  Lets tests pin the kernels' geometry bit for bit.
 */
ORC_API void orc_geometry(const float cam36[36], const uint32_t dims[3], uint32_t W, uint32_t H, float *out8)
{
    uint32_t d4[4];
    float bmin[4], bsz[4], buf[4];
    orc_grid_uniform(dims, 1.0f, d4, bmin, bsz, buf);
    float bmax[3] = { bmin[0] + bsz[0], bmin[1] + bsz[1], bmin[2] + bsz[2] };
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t py = 0; py < (int64_t)H; ++py) {
        for (uint32_t px = 0; px < W; ++px) {
            float rd[3], t_min, t_max;
            int hit = orc_ray(cam36, dims, px, (uint32_t)py, W, H, rd, &t_min, &t_max);
            float *o = out8 + 8 * ((size_t)py * W + px);
            uint32_t n_in = 0, iters = 0;
            if (hit) {
                float dt = t_min;
                for (int i = 0; i < MAX_STEP_AMOUNT; ++i) {
                    ++iters;
                    float p[3] = { cam36[0] + rd[0] * dt, cam36[1] + rd[1] * dt, cam36[2] + rd[2] * dt };
                    int in = !(p[0] < bmin[0] || p[1] < bmin[1] || p[2] < bmin[2]) &&
                             !(p[0] >= bmax[0] || p[1] >= bmax[1] || p[2] >= bmax[2]);
                    n_in += (uint32_t)in;
                    dt += STEP_SIZE;
                    if (dt >= t_max) break;
                }
            }
            o[0] = rd[0]; o[1] = rd[1]; o[2] = rd[2]; o[3] = hit ? t_min : 0.0f;
            o[4] = hit ? t_max : 0.0f; o[5] = (float)n_in; o[6] = hit ? 1.0f : 0.0f; o[7] = (float)iters;
        }
    }
}

/* ---- compare/netcdf.rs:8-83: voxel-space RMSE between a grid and a color(c,z,y,x) f32 volume ------------------
 * Loop order (x outer, y, z inner), the single running f32 sum, powi(.,2) as a product, remap() and mix() with the
 * reference's operation order, the mean over x*y*z*amount_channels and the final f32 sqrt are all the reference's.
 * tf9 = color_a.rgb, color_b.rgb, color_c.rgb (grid.rs:285-287).  *out_f64 (may be NULL) receives the same
 * quantity with the sum carried in f64, which is what the GPU reduction computes. */
static float cmp_remap(float value, float min1, float max1, float min2, float max2) /* netcdf.rs:81-83 */
{
    return min2 + (value - min1) * (max2 - min2) / (max1 - min1);
}
static float cmp_mix(float x, float y, float a) { return x * (1.0f - a) + y * a; } /* netcdf.rs:77-79 */

ORC_API float orc_compare_volume(const uint8_t *voxels, const float *data, uint32_t X, uint32_t Y, uint32_t Z,
                                 const float tf9[9], int density_only, double *out_f64)
{
    const size_t n = (size_t)X * Y * Z;
    float rmse = 0.0f;
    double sum64 = 0.0;
    int amount_channels = 1;
    for (uint32_t x = 0; x < X; ++x)
        for (uint32_t y = 0; y < Y; ++y)
            for (uint32_t z = 0; z < Z; ++z) {
                const size_t i = ((size_t)z * Y + y) * X + x;                 /* grid.rs:211, data[[c,z,y,x]] */
                const float alpha = (float)voxels[4 * i + 3] / 255.0f;        /* netcdf.rs:31 */
                float t = alpha - data[3 * n + i];
                t = t * t;
                rmse += t;                                                     /* netcdf.rs:32 */
                sum64 += (double)t;
                if (!density_only) {
                    float s[3], e[3], ar;
                    if (alpha < 0.5f) {
                        for (int k = 0; k < 3; ++k) { s[k] = tf9[k]; e[k] = tf9[3 + k]; }
                        ar = cmp_remap(alpha, 0.0f, 0.5f, 0.0f, 1.0f);
                    } else {
                        for (int k = 0; k < 3; ++k) { s[k] = tf9[3 + k]; e[k] = tf9[6 + k]; }
                        ar = cmp_remap(alpha, 0.5f, 1.0f, 0.0f, 1.0f);
                    }
                    for (int k = 0; k < 3; ++k) {
                        float d = cmp_mix(s[k], e[k], ar) - data[(size_t)k * n + i];
                        d = d * d;
                        rmse += d;                                             /* netcdf.rs:62-64 */
                        sum64 += (double)d;
                    }
                    amount_channels = 4;
                }
            }
    if (out_f64) *out_f64 = sqrt(sum64 / ((double)n * (double)amount_channels));
    const float mean = rmse / (float)(n * (size_t)amount_channels);           /* netcdf.rs:73 */
    return sqrtf(mean);
}

ORC_API int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
