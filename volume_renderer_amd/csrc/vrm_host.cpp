// vrm_host.cpp -- host side of the path above the kernels: camera matrices, sphere view list,
// PNG + cameras.json writers, DAT loader and the multi-view export loop.
//
// The reference keeps these in Rust (camera.rs, transform.rs, camera_sphere_controller.rs,
// sphere_screenshot_manager.rs, screenshot.rs, loaders/dat.rs) on top of glam / image / serde_json.
// Those crates are not vendored; the formulas below follow their published definitions.
#include <sys/stat.h>
#include <zlib.h>

#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "vrm_internal.h"

namespace {

// ---- minimal f32 linear algebra in glam's conventions (column-major, right-handed) -----------
struct V3 {
    float x, y, z;
};
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V3 normalize(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

struct Q {
    float x, y, z, w;
};
struct M4 {
    float m[16]; // m[4*col + row]
    float &at(int col, int row) { return m[4 * col + row]; }
    float at(int col, int row) const { return m[4 * col + row]; }
};

// glam Mat4::look_to_rh
M4 look_to_rh(V3 eye, V3 dir, V3 up)
{
    const V3 f = normalize(dir), s = normalize(cross(f, up)), u = cross(s, f);
    M4 r{};
    r.at(0, 0) = s.x; r.at(0, 1) = u.x; r.at(0, 2) = -f.x;
    r.at(1, 0) = s.y; r.at(1, 1) = u.y; r.at(1, 2) = -f.y;
    r.at(2, 0) = s.z; r.at(2, 1) = u.z; r.at(2, 2) = -f.z;
    r.at(3, 0) = -dot(eye, s); r.at(3, 1) = -dot(eye, u); r.at(3, 2) = dot(eye, f); r.at(3, 3) = 1.0f;
    return r;
}

// glam Mat4::perspective_rh (depth 0..1)
M4 perspective_rh(float fovy, float aspect, float z_near, float z_far)
{
    const float sn = std::sin(0.5f * fovy), cs = std::cos(0.5f * fovy);
    const float h = cs / sn, w = h / aspect, r = z_far / (z_near - z_far);
    M4 p{};
    p.at(0, 0) = w;
    p.at(1, 1) = h;
    p.at(2, 2) = r; p.at(2, 3) = -1.0f;
    p.at(3, 2) = r * z_near;
    return p;
}

M4 mul(const M4 &a, const M4 &b)
{
    M4 r{};
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 4; ++i) {
            float acc = a.at(0, i) * b.at(c, 0);
            for (int k = 1; k < 4; ++k) acc += a.at(k, i) * b.at(c, k);
            r.at(c, i) = acc;
        }
    return r;
}

// 4x4 inverse in the operation order of glam's Mat4::inverse (the GLM cofactor scheme: six vectors of
// 2x2 sub-determinants, four signed column products, determinant from the first row).  VP with far =
// 1000 is ill-conditioned (entries ~650 that cancel to ~0.6 in the shader's M*v), so the order matters
// at the 1e-4 level and is kept as close to the reference's library as its published algorithm allows.
struct V4 {
    float v[4];
};
inline V4 vmul(V4 a, V4 b) { return {{a.v[0] * b.v[0], a.v[1] * b.v[1], a.v[2] * b.v[2], a.v[3] * b.v[3]}}; }
inline V4 vsub(V4 a, V4 b) { return {{a.v[0] - b.v[0], a.v[1] - b.v[1], a.v[2] - b.v[2], a.v[3] - b.v[3]}}; }
inline V4 vadd(V4 a, V4 b) { return {{a.v[0] + b.v[0], a.v[1] + b.v[1], a.v[2] + b.v[2], a.v[3] + b.v[3]}}; }

M4 inverse(const M4 &a)
{
    auto sub = [&](int c0, int r0, int c1, int r1, int c2, int r2, int c3, int r3) {
        return a.at(c0, r0) * a.at(c1, r1) - a.at(c2, r2) * a.at(c3, r3);
    };
    const float k00 = sub(2, 2, 3, 3, 3, 2, 2, 3), k02 = sub(1, 2, 3, 3, 3, 2, 1, 3), k03 = sub(1, 2, 2, 3, 2, 2, 1, 3);
    const float k04 = sub(2, 1, 3, 3, 3, 1, 2, 3), k06 = sub(1, 1, 3, 3, 3, 1, 1, 3), k07 = sub(1, 1, 2, 3, 2, 1, 1, 3);
    const float k08 = sub(2, 1, 3, 2, 3, 1, 2, 2), k10 = sub(1, 1, 3, 2, 3, 1, 1, 2), k11 = sub(1, 1, 2, 2, 2, 1, 1, 2);
    const float k12 = sub(2, 0, 3, 3, 3, 0, 2, 3), k14 = sub(1, 0, 3, 3, 3, 0, 1, 3), k15 = sub(1, 0, 2, 3, 2, 0, 1, 3);
    const float k16 = sub(2, 0, 3, 2, 3, 0, 2, 2), k18 = sub(1, 0, 3, 2, 3, 0, 1, 2), k19 = sub(1, 0, 2, 2, 2, 0, 1, 2);
    const float k20 = sub(2, 0, 3, 1, 3, 0, 2, 1), k22 = sub(1, 0, 3, 1, 3, 0, 1, 1), k23 = sub(1, 0, 2, 1, 2, 0, 1, 1);
    const V4 f0{{k00, k00, k02, k03}}, f1{{k04, k04, k06, k07}}, f2{{k08, k08, k10, k11}};
    const V4 f3{{k12, k12, k14, k15}}, f4{{k16, k16, k18, k19}}, f5{{k20, k20, k22, k23}};
    const V4 r0{{a.at(1, 0), a.at(0, 0), a.at(0, 0), a.at(0, 0)}}, r1{{a.at(1, 1), a.at(0, 1), a.at(0, 1), a.at(0, 1)}};
    const V4 r2{{a.at(1, 2), a.at(0, 2), a.at(0, 2), a.at(0, 2)}}, r3{{a.at(1, 3), a.at(0, 3), a.at(0, 3), a.at(0, 3)}};
    const V4 sp{{1.0f, -1.0f, 1.0f, -1.0f}}, sn{{-1.0f, 1.0f, -1.0f, 1.0f}};
    const V4 c0 = vmul(vadd(vsub(vmul(r1, f0), vmul(r2, f1)), vmul(r3, f2)), sp);
    const V4 c1 = vmul(vadd(vsub(vmul(r0, f0), vmul(r2, f3)), vmul(r3, f4)), sn);
    const V4 c2 = vmul(vadd(vsub(vmul(r0, f1), vmul(r1, f3)), vmul(r3, f5)), sp);
    const V4 c3 = vmul(vadd(vsub(vmul(r0, f2), vmul(r1, f4)), vmul(r2, f5)), sn);
    const float det = ((a.at(0, 0) * c0.v[0] + a.at(0, 1) * c1.v[0]) + a.at(0, 2) * c2.v[0]) + a.at(0, 3) * c3.v[0];
    const float rcp = 1.0f / det;
    M4 inv{};
    for (int i = 0; i < 4; ++i) {
        inv.at(0, i) = c0.v[i] * rcp;
        inv.at(1, i) = c1.v[i] * rcp;
        inv.at(2, i) = c2.v[i] * rcp;
        inv.at(3, i) = c3.v[i] * rcp;
    }
    return inv;
}

// glam Quat::from_mat4 (rotation part; M. Day's branch scheme)
Q quat_from_rotation(const M4 &m)
{
    const float m00 = m.at(0, 0), m01 = m.at(0, 1), m02 = m.at(0, 2);
    const float m10 = m.at(1, 0), m11 = m.at(1, 1), m12 = m.at(1, 2);
    const float m20 = m.at(2, 0), m21 = m.at(2, 1), m22 = m.at(2, 2);
    if (m22 <= 0.0f) {
        const float d = m11 - m00, o = 1.0f - m22;
        if (d <= 0.0f) {
            const float s = o - d, k = 0.5f / std::sqrt(s);
            return {s * k, (m01 + m10) * k, (m02 + m20) * k, (m12 - m21) * k};
        }
        const float s = o + d, k = 0.5f / std::sqrt(s);
        return {(m01 + m10) * k, s * k, (m12 + m21) * k, (m20 - m02) * k};
    }
    const float d = m11 + m00, o = 1.0f + m22;
    if (d <= 0.0f) {
        const float s = o - d, k = 0.5f / std::sqrt(s);
        return {(m02 + m20) * k, (m12 + m21) * k, s * k, (m01 - m10) * k};
    }
    const float s = o + d, k = 0.5f / std::sqrt(s);
    return {(m12 - m21) * k, (m20 - m02) * k, (m01 - m10) * k, s * k};
}

// glam Quat * Vec3
V3 rotate(Q q, V3 v)
{
    const V3 b{q.x, q.y, q.z};
    const float b2 = dot(b, b);
    return v * (q.w * q.w - b2) + b * (dot(v, b) * 2.0f) + cross(b, v) * (q.w * 2.0f);
}

struct CameraBasis {
    V3 right, up, front;
};

// camera.rs:26-31,61-67 + transform.rs:30-32,59-63
void camera_uniform(V3 pos, V3 target, float aspect, vrm_camera_uniform *out, CameraBasis *basis)
{
    const V3 up{0.0f, 1.0f, 0.0f};
    const V3 d = normalize(target - pos);
    Q rot = quat_from_rotation(look_to_rh(pos, d, up));
    rot.x = -rot.x; rot.y = -rot.y; rot.z = -rot.z; // conjugate
    const V3 forward = rotate(rot, V3{0.0f, 0.0f, -1.0f});
    const M4 view = look_to_rh(pos, forward, up);
    const float fovy = 45.0f * (3.14159265358979323846f / 180.0f);
    const M4 proj = perspective_rh(fovy, aspect, 0.1f, 1000.0f);
    const M4 vp = mul(proj, view);
    const M4 ivp = inverse(vp);
    out->position[0] = pos.x; out->position[1] = pos.y; out->position[2] = pos.z; out->position[3] = 0.0f;
    memcpy(out->view_proj, vp.m, 64);
    memcpy(out->inverse_view_proj, ivp.m, 64);
    if (basis) {
        basis->right = rotate(rot, V3{1.0f, 0.0f, 0.0f});
        basis->up = rotate(rot, V3{0.0f, 1.0f, 0.0f});
        basis->front = forward;
    }
}

// camera_sphere_controller.rs:79-89
V3 sphere_position(uint32_t ix, uint32_t iy, uint32_t xdiv, uint32_t ydiv, float radius, V3 origin)
{
    const float PI = 3.14159265358979323846f;
    const float theta = (float)ix / (float)xdiv * PI * 2.0f;
    const float phi = (float)iy / (float)ydiv * PI;
    const float x = std::sin(phi) * std::cos(theta) * radius;
    const float y = std::sin(phi) * std::sin(theta) * radius;
    const float z = std::cos(phi) * radius;
    return V3{x, z, -y} + origin;
}

// ---- PNG -------------------------------------------------------------------------------------------
void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

void png_chunk(std::vector<uint8_t> &out, const char type[4], const uint8_t *data, size_t n)
{
    put_be32(out, (uint32_t)n);
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    put_be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(n + 4)));
}

// shortest round-trip f32, JSON-safe, in the style serde_json/ryu prints ("45.0", "1e-7")
std::string json_f32(float v)
{
    if (!std::isfinite(v)) return "null"; // serde_json writes null for non-finite floats
    char buf[48];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    std::string s(buf, r.ptr);
    const size_t e = s.find('e');
    if (e == std::string::npos) {
        if (s.find('.') == std::string::npos) s += ".0";
    } else {
        std::string mant = s.substr(0, e), ex = s.substr(e + 1);
        bool neg = false;
        if (!ex.empty() && (ex[0] == '+' || ex[0] == '-')) { neg = ex[0] == '-'; ex.erase(0, 1); }
        while (ex.size() > 1 && ex[0] == '0') ex.erase(0, 1);
        s = mant + "e" + (neg ? "-" : "") + ex;
    }
    return s;
}

void json_vec3(std::ostream &os, const char *indent, const char *key, V3 v, bool comma)
{
    os << indent << "\"" << key << "\": [\n";
    os << indent << "  " << json_f32(v.x) << ",\n" << indent << "  " << json_f32(v.y) << ",\n" << indent << "  " << json_f32(v.z) << "\n";
    os << indent << "]" << (comma ? "," : "") << "\n";
}

} // namespace

int vrm_write_png_rgba8(const char *path, const uint8_t *rgba, uint32_t w, uint32_t h, size_t pitch)
{
    // filter type 0 on every scanline, one zlib stream (image-rs writes the same pixel data; the
    // compressed bytes are encoder-specific, the decoded image is what parity is defined on)
    std::vector<uint8_t> raw((size_t)h * ((size_t)w * 4 + 1));
    for (uint32_t y = 0; y < h; ++y) {
        uint8_t *row = raw.data() + (size_t)y * ((size_t)w * 4 + 1);
        row[0] = 0;
        memcpy(row + 1, rgba + (size_t)y * pitch, (size_t)w * 4);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 1) != Z_OK) {
        vrm_set_error("zlib compress2 failed for %s", path);
        return VRM_ERR_IO;
    }
    std::vector<uint8_t> out;
    out.reserve(clen + 64);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    const uint8_t tail[5] = {8, 6, 0, 0, 0}; // 8-bit, colour type 6 (RGBA), deflate, adaptive, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    png_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    png_chunk(out, "IDAT", comp.data(), clen);
    png_chunk(out, "IEND", nullptr, 0);
    FILE *f = fopen(path, "wb");
    if (!f) {
        vrm_set_error("cannot open %s for writing", path);
        return VRM_ERR_IO;
    }
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok) {
        vrm_set_error("short write to %s", path);
        return VRM_ERR_IO;
    }
    return VRM_OK;
}

extern "C" int vrm_camera_look_at(const float position[3], const float target[3], float aspect, vrm_camera_uniform *out,
                                  float basis9[9])
{
    if (!position || !target || !out) {
        vrm_set_error("vrm_camera_look_at: null argument");
        return VRM_ERR_INVALID;
    }
    CameraBasis b;
    camera_uniform(V3{position[0], position[1], position[2]}, V3{target[0], target[1], target[2]}, aspect, out, &b);
    if (basis9) {
        const float v[9] = {b.right.x, b.right.y, b.right.z, b.up.x, b.up.y, b.up.z, b.front.x, b.front.y, b.front.z};
        memcpy(basis9, v, sizeof v);
    }
    return VRM_OK;
}

extern "C" int vrm_sphere_position(uint32_t ix, uint32_t iy, uint32_t x_div, uint32_t y_div, float radius, const float origin[3],
                                   float out_position[3])
{
    if (!origin || !out_position || x_div == 0 || y_div == 0) {
        vrm_set_error("vrm_sphere_position: bad argument");
        return VRM_ERR_INVALID;
    }
    const V3 p = sphere_position(ix, iy, x_div, y_div, radius, V3{origin[0], origin[1], origin[2]});
    out_position[0] = p.x; out_position[1] = p.y; out_position[2] = p.z;
    return VRM_OK;
}

// "Screenshot All": sphere_screenshot_manager.rs:23-73 + state.rs:391-397, without the one-view-per-
// presented-frame blocking: views are rendered in batches, read back asynchronously and PNG-encoded
// on host threads while the next batch renders.
extern "C" int vrm_export_sphere(vrm_ctx *ctx, uint32_t x_div, uint32_t y_div, float radius, const float origin[3],
                                 const char *out_dir, uint32_t part, uint32_t n_parts)
{
    return vrm_guard("vrm_export_sphere", VRM_ERR_IO, [&]() -> int {
    if (!ctx || !origin || !out_dir || x_div == 0 || y_div < 2 || n_parts == 0 || part >= n_parts) {
        vrm_set_error("vrm_export_sphere: bad argument (x_div >= 1, y_div >= 2, part < n_parts)");
        return VRM_ERR_INVALID;
    }
    struct View {
        uint32_t ix, iy;
        vrm_camera_uniform cam;
        CameraBasis basis;
        V3 pos;
    };
    const V3 org{origin[0], origin[1], origin[2]};
    const float aspect = (float)ctx->def_width / (float)ctx->def_height; // state.rs:104
    std::vector<View> all;
    for (uint32_t iy = 1; iy < y_div; ++iy)      // ssm.rs:25-26,36-41: y outer from 1, x inner
        for (uint32_t ix = 0; ix < x_div; ++ix) {
            View v;
            v.ix = ix; v.iy = iy;
            v.pos = sphere_position(ix, iy, x_div, y_div, radius, org);
            camera_uniform(v.pos, org, aspect, &v.cam, &v.basis);
            all.push_back(v);
        }
    const std::string dir(out_dir);

    if (part == 0) { // ssm.rs:68-73 + struct layout ssm.rs:78-93 (serde_json::to_writer_pretty)
        std::ofstream js(dir + "/cameras.json");
        if (!js) {
            vrm_set_error("cannot write %s/cameras.json", out_dir);
            return VRM_ERR_IO;
        }
        js << "{\n";
        json_vec3(js, "  ", "look_at", org, true);
        js << "  \"positions\": [\n";
        for (size_t i = 0; i < all.size(); ++i) {
            const View &v = all[i];
            js << "    {\n";
            json_vec3(js, "      ", "position", v.pos, true);
            json_vec3(js, "      ", "right", v.basis.right, true);
            json_vec3(js, "      ", "up", v.basis.up, true);
            json_vec3(js, "      ", "front", v.basis.front, true);
            js << "      \"img\": \"" << v.ix << "-" << v.iy << ".png\",\n";
            js << "      \"fovy\": " << json_f32(45.0f) << "\n";
            js << "    }" << (i + 1 < all.size() ? "," : "") << "\n";
        }
        js << "  ]\n}";
        if (!js.good()) {
            vrm_set_error("short write to %s/cameras.json", out_dir);
            return VRM_ERR_IO;
        }
    }

    std::vector<const View *> mine;
    for (size_t i = 0; i < all.size(); ++i)
        if ((i + i / x_div) % n_parts == part) mine.push_back(&all[i]); // interleaved, rotated by one per row of the sphere (see vrm.h)
    if (mine.empty()) return VRM_OK;

    const uint32_t bank = (uint32_t)std::min<size_t>(8, mine.size());
    std::vector<vrm_image *> imgs(2 * bank, nullptr);
    int rc = VRM_OK;
    for (auto &im : imgs)
        if (!rc) rc = vrm_image_create(ctx, ctx->def_width, ctx->def_height, &im);

    auto issue = [&](size_t first, uint32_t which) -> int { // draw + async readback of up to `bank` views
        const size_t n = std::min<size_t>(bank, mine.size() - first);
        std::vector<vrm_camera_uniform> cams(n);
        for (size_t i = 0; i < n; ++i) cams[i] = mine[first + i]->cam;
        int r = vrm_draw_batch(ctx, cams.data(), (uint32_t)n, imgs.data() + which * bank);
        for (size_t i = 0; i < n && !r; ++i) r = vrm_image_readback_async(imgs[which * bank + i]);
        return r;
    };
    auto encode = [&](size_t first, uint32_t which) -> int { // PNG-encode a finished bank on host threads
        const size_t n = std::min<size_t>(bank, mine.size() - first);
        std::vector<int> res(n, VRM_OK);
        std::vector<std::string> msgs(n);
        std::vector<std::thread> th;
        size_t mapped = 0;
        int r = VRM_OK;
        std::string first_msg;
        // A failure part-way (a HIP error from a map, no thread to be had) must not leave joinable threads behind --
        // their destructor would terminate the process -- nor threads still reading pinned buffers the caller is
        // about to free: stop starting work, join what runs, unmap what is mapped, then report.
        try {
            for (size_t i = 0; i < n && !r; ++i) {
                vrm_image *im = imgs[which * bank + i];
                const uint8_t *pxp = nullptr;
                size_t pitch = 0;
                r = vrm_image_map(im, &pxp, &pitch); // blocks until this image's copy landed
                if (r) {
                    first_msg = vrm_last_error();
                    break;
                }
                ++mapped;
                const View *v = mine[first + i];
                std::string path = dir + "/" + std::to_string(v->ix) + "-" + std::to_string(v->iy) + ".png"; // state.rs:393
                th.emplace_back([=, &res, &msgs]() {
                    res[i] = vrm_write_png_rgba8(path.c_str(), pxp, im->width, im->height, pitch);
                    if (res[i]) msgs[i] = vrm_last_error();
                });
            }
        } catch (const std::exception &e) {
            r = VRM_ERR_IO;
            first_msg = std::string("export: ") + e.what();
        }
        for (auto &t : th) t.join();
        for (size_t i = 0; i < mapped; ++i) (void)vrm_image_unmap(imgs[which * bank + i]);
        for (size_t i = 0; i < n && !r; ++i)
            if (res[i]) {
                r = res[i];
                first_msg = msgs[i];
            }
        if (r) vrm_set_error("%s", first_msg.c_str());
        return r;
    };

    if (!rc) rc = issue(0, 0);
    uint32_t which = 0;
    for (size_t first = 0; first < mine.size() && !rc; first += bank) {
        const size_t next = first + bank;
        if (next < mine.size()) rc = issue(next, which ^ 1u); // GPU renders the next bank while we encode this one
        if (!rc) rc = encode(first, which);
        which ^= 1u;
    }
    std::string keep = rc ? vrm_last_error() : "";
    (void)vrm_synchronize(ctx);
    for (auto im : imgs) vrm_image_destroy(im);
    if (rc) vrm_set_error("%s", keep.c_str());
    return rc;
    });
}

// loaders/dat.rs:10-82
extern "C" int vrm_load_dat(vrm_ctx *ctx, const char *dat_path)
{
    return vrm_guard("vrm_load_dat", VRM_ERR_IO, [&]() -> int {
    if (!ctx || !dat_path) {
        vrm_set_error("vrm_load_dat: null argument");
        return VRM_ERR_INVALID;
    }
    std::ifstream in(dat_path);
    if (!in) {
        vrm_set_error("cannot open %s", dat_path);
        return VRM_ERR_IO;
    }
    auto trim = [](std::string s) {
        const char *ws = " \t\r\n";
        const size_t a = s.find_first_not_of(ws);
        if (a == std::string::npos) return std::string();
        return s.substr(a, s.find_last_not_of(ws) - a + 1);
    };
    std::string line, raw_name;
    uint32_t res[3] = {0, 0, 0};
    bool have_res = false;
    uint32_t bps = 1;
    while (std::getline(in, line)) {
        const size_t c = line.find(':');
        if (c == std::string::npos) continue; // the reference would panic here (dat.rs:20); we skip the line
        const std::string key = line.substr(0, c);
        std::string rest = line.substr(c + 1);
        const size_t c2 = rest.find(':'); // split(":")[1]: up to the next colon
        if (c2 != std::string::npos) rest = rest.substr(0, c2);
        const std::string value = trim(rest);
        if (key == "ObjectFileName") raw_name = value;
        else if (key == "Resolution") {
            std::istringstream ss(value);
            if (!(ss >> res[0] >> res[1] >> res[2])) {
                vrm_set_error("%s: cannot parse Resolution '%s'", dat_path, value.c_str());
                return VRM_ERR_IO;
            }
            have_res = true;
        } else if (key == "Format") {
            if (value == "USHORT") bps = 2; // anything else is read as one byte per sample (dat.rs:32-37)
        }
    }
    if (raw_name.empty() || !have_res) {
        vrm_set_error("%s: missing ObjectFileName or Resolution", dat_path);
        return VRM_ERR_IO;
    }
    std::string dirp(dat_path);
    const size_t slash = dirp.find_last_of('/');
    dirp = (slash == std::string::npos) ? std::string(".") : dirp.substr(0, slash);
    const std::string raw_path = dirp + "/" + raw_name;
    FILE *f = fopen(raw_path.c_str(), "rb");
    if (!f) {
        vrm_set_error("cannot open %s", raw_path.c_str());
        return VRM_ERR_IO;
    }
    const size_t need = (size_t)res[0] * res[1] * res[2] * bps;
    // the header must not size an allocation the file cannot fill
    struct stat st;
    if (fstat(fileno(f), &st) != 0 || need == 0 || (uint64_t)st.st_size < (uint64_t)need ||
        (uint64_t)res[0] * res[1] > ((uint64_t)1 << 40) / res[2]) {
        fclose(f);
        vrm_set_error("%s: expected %zu bytes of samples (%u x %u x %u x %u), file has %lld", raw_path.c_str(), need, res[0], res[1],
                      res[2], bps, (long long)st.st_size);
        return VRM_ERR_IO;
    }
    std::vector<uint8_t> raw(need);
    const size_t got = need ? fread(raw.data(), 1, need, f) : 0;
    fclose(f);
    if (got != need || need == 0) {
        vrm_set_error("%s: expected %zu bytes of samples, file has %zu", raw_path.c_str(), need, got);
        return VRM_ERR_IO;
    }
    int rc = vrm_grid_resize(ctx, res[0], res[1], res[2]); // dat.rs:57
    if (rc) return rc;
    return vrm_grid_upload_scalar(ctx, raw.data(), raw.size(), bps);
    });
}
