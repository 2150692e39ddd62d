// vrm_eval.cpp -- evaluation helpers either side of the ray-marching path: the voxel-space
// RMSE of compare/netcdf.rs and the per-image metric of scripts/similarity.py, as GPU reductions behind the C ABI,
// plus the PNG reader the image-metric CLI needs.  HIP runtime + zlib only.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "vrm_internal.h"

#define HIP_TRY(call)                                     \
    do {                                                  \
        int rc_ = vrm_check_hip((call), #call);           \
        if (rc_ != VRM_OK) return rc_;                    \
    } while (0)

#define REQUIRE(cond, ...)                                \
    do {                                                  \
        if (!(cond)) {                                    \
            vrm_set_error(__VA_ARGS__);                   \
            return VRM_ERR_INVALID;                       \
        }                                                 \
    } while (0)

namespace {

struct DevBuf { // freed on every return path
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

int compare_volume(vrm_ctx *ctx, const float *data_dev, int density_only, float *out_rmse)
{
    const size_t n = ctx->vox_words;
    if (int rc_ = vrm_grid_expand_internal(ctx)) return rc_; // (the compare reads the x-fastest RGBA8 voxels)
    const uint32_t blocks = vrm_reduce_blocks(n);
    DevBuf partial;
    HIP_TRY(hipMalloc(&partial.p, blocks * sizeof(double)));
    const float tf9[9] = {ctx->tf.color_a[0], ctx->tf.color_a[1], ctx->tf.color_a[2], ctx->tf.color_b[0], ctx->tf.color_b[1],
                          ctx->tf.color_b[2], ctx->tf.color_c[0], ctx->tf.color_c[1], ctx->tf.color_c[2]};
    HIP_TRY(vrm_launch_compare_volume(ctx->vox, data_dev, n, density_only != 0, tf9, (double *)partial.p, blocks, ctx->stream));
    std::vector<double> h(blocks);
    HIP_TRY(hipMemcpyAsync(h.data(), partial.p, blocks * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double sum = 0.0;
    for (double v : h) sum += v; // fixed order: the result is deterministic
    // netcdf.rs:73-74: mean over x*y*z*amount_channels, then the square root
    const double mean = sum / ((double)n * (density_only ? 1.0 : 4.0));
    *out_rmse = (float)std::sqrt(mean);
    return VRM_OK;
}

int ssd_device(vrm_ctx *ctx, const uint8_t *a, const uint8_t *b, size_t row_words, size_t rows, size_t pitch_a, size_t pitch_b,
               uint64_t *out)
{
    const uint32_t blocks = vrm_reduce_blocks(row_words * rows);
    DevBuf partial;
    HIP_TRY(hipMalloc(&partial.p, blocks * sizeof(unsigned long long)));
    HIP_TRY(vrm_launch_ssd_u8(a, b, row_words, rows, pitch_a, pitch_b, (unsigned long long *)partial.p, blocks, ctx->stream));
    std::vector<unsigned long long> h(blocks);
    HIP_TRY(hipMemcpyAsync(h.data(), partial.p, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    uint64_t s = 0;
    for (auto v : h) s += v;
    *out = s;
    return VRM_OK;
}

} // namespace

extern "C" int vrm_compare_volume_f32_device(vrm_ctx *ctx, const float *color_czyx_dev, int density_only, float *out_rmse)
{
    return vrm_guard("vrm_compare_volume_f32_device", VRM_ERR_INVALID, [&]() -> int {
    REQUIRE(ctx != nullptr && color_czyx_dev != nullptr && out_rmse != nullptr, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    return compare_volume(ctx, color_czyx_dev, density_only, out_rmse);
    });
}

extern "C" int vrm_compare_volume_f32(vrm_ctx *ctx, const float *color_czyx, int density_only, float *out_rmse)
{
    return vrm_guard("vrm_compare_volume_f32", VRM_ERR_INVALID, [&]() -> int {
    REQUIRE(ctx != nullptr && color_czyx != nullptr && out_rmse != nullptr, "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    DevBuf d;
    const size_t bytes = ctx->vox_words * 4 * sizeof(float);
    HIP_TRY(hipMalloc(&d.p, bytes));
    HIP_TRY(hipMemcpyAsync(d.p, color_czyx, bytes, hipMemcpyHostToDevice, ctx->stream));
    return compare_volume(ctx, (const float *)d.p, density_only, out_rmse);
    });
}

extern "C" int vrm_ssd_u8(vrm_ctx *ctx, const uint8_t *a, const uint8_t *b, size_t n, uint64_t *out_ssd)
{
    return vrm_guard("vrm_ssd_u8", VRM_ERR_INVALID, [&]() -> int {
    REQUIRE(ctx != nullptr && out_ssd != nullptr && (n == 0 || (a != nullptr && b != nullptr)), "null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    *out_ssd = 0;
    if (n == 0) return VRM_OK;
    const size_t padded = (n + 3) & ~(size_t)3; // the tail words compare zero padding with zero padding
    DevBuf da, db;
    HIP_TRY(hipMalloc(&da.p, padded));
    HIP_TRY(hipMalloc(&db.p, padded));
    HIP_TRY(hipMemsetAsync((uint8_t *)da.p + (padded - 4), 0, 4, ctx->stream));
    HIP_TRY(hipMemsetAsync((uint8_t *)db.p + (padded - 4), 0, 4, ctx->stream));
    HIP_TRY(hipMemcpyAsync(da.p, a, n, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db.p, b, n, hipMemcpyHostToDevice, ctx->stream));
    return ssd_device(ctx, (const uint8_t *)da.p, (const uint8_t *)db.p, padded / 4, 1, padded, padded, out_ssd);
    });
}

extern "C" int vrm_ssim_u8(vrm_ctx *ctx, const uint8_t *a, const uint8_t *b, uint32_t width, uint32_t height, uint32_t channels,
                           double data_range, double *out_ssim)
{
    return vrm_guard("vrm_ssim_u8", VRM_ERR_INVALID, [&]() -> int {
    REQUIRE(ctx != nullptr && a != nullptr && b != nullptr && out_ssim != nullptr, "null argument");
    REQUIRE(channels >= 1 && channels <= 4, "channels must be 1..4, got %u", channels);
    // skimage: "win_size exceeds image extent" (win_size = 7)
    REQUIRE(width >= 7 && height >= 7, "image %ux%u is smaller than the 7x7 SSIM window", width, height);
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)width * height * channels;
    DevBuf da, db, partial;
    HIP_TRY(hipMalloc(&da.p, bytes));
    HIP_TRY(hipMalloc(&db.p, bytes));
    HIP_TRY(hipMemcpyAsync(da.p, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db.p, b, bytes, hipMemcpyHostToDevice, ctx->stream));
    const size_t items = (size_t)(width - 6) * (height - 6) * channels;
    const uint32_t blocks = vrm_reduce_blocks(items);
    HIP_TRY(hipMalloc(&partial.p, blocks * sizeof(double)));
    const double c1 = (0.01 * data_range) * (0.01 * data_range), c2 = (0.03 * data_range) * (0.03 * data_range);
    HIP_TRY(vrm_launch_ssim_u8((const uint8_t *)da.p, (const uint8_t *)db.p, width, height, channels, c1, c2, (double *)partial.p, blocks,
                               ctx->stream));
    std::vector<double> h(blocks);
    HIP_TRY(hipMemcpyAsync(h.data(), partial.p, blocks * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    double sum = 0.0;
    for (double v : h) sum += v; // fixed order: deterministic
    // per-channel means over the cropped map, then the mean over channels: every channel has the same count
    *out_ssim = sum / (double)items;
    return VRM_OK;
    });
}

extern "C" int vrm_image_ssd(vrm_image *a, vrm_image *b, uint64_t *out_ssd, uint64_t *out_count)
{
    return vrm_guard("vrm_image_ssd", VRM_ERR_INVALID, [&]() -> int {
    REQUIRE(a != nullptr && b != nullptr && out_ssd != nullptr, "null argument");
    REQUIRE(a->ctx != nullptr && a->ctx == b->ctx, "images belong to different contexts or their context was destroyed");
    REQUIRE(a->width == b->width && a->height == b->height, "image sizes differ (%ux%u vs %ux%u)", a->width, a->height, b->width,
            b->height);
    vrm_ctx *ctx = a->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    if (out_count) *out_count = (uint64_t)a->width * a->height * 4;
    // both images were drawn on ctx->stream, the reduction is enqueued behind them
    return ssd_device(ctx, (const uint8_t *)a->dev, (const uint8_t *)b->dev, a->width, a->height, a->pitch, b->pitch, out_ssd);
    });
}

// ---- PNG reader: 8-bit gray / gray+alpha / RGB / RGBA, non-interlaced (what the exporter and DiffDVR write) ----

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

extern "C" int vrm_png_read(const char *path, uint32_t *width, uint32_t *height, uint32_t *channels, uint8_t *pixels, size_t capacity)
{
    return vrm_guard("vrm_png_read", VRM_ERR_IO, [&]() -> int {
    REQUIRE(path != nullptr && width != nullptr && height != nullptr && channels != nullptr, "null argument");
    FILE *f = fopen(path, "rb");
    if (!f) {
        vrm_set_error("cannot open %s", path);
        return VRM_ERR_IO;
    }
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + got);
    fclose(f);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) {
        vrm_set_error("%s is not a PNG file", path);
        return VRM_ERR_IO;
    }
    uint32_t w = 0, h = 0, ch = 0;
    std::vector<uint8_t> idat;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t *type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) break;
        const uint8_t *data = &file[pos + 8];
        if (be32(&file[pos + 8 + len]) != (uint32_t)crc32(crc32(0L, type, 4), data, len)) {
            vrm_set_error("%s: chunk CRC mismatch", path);
            return VRM_ERR_IO;
        }
        if (!memcmp(type, "IHDR", 4) && len == 13) {
            w = be32(data);
            h = be32(data + 4);
            const int depth = data[8], ctype = data[9], interlace = data[12];
            ch = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
            if (depth != 8 || ch == 0 || interlace != 0) {
                vrm_set_error("%s: only 8-bit non-interlaced gray/RGB(A) PNGs are supported (depth %d, colour type %d, interlace %d)",
                              path, depth, ctype, interlace);
                return VRM_ERR_IO;
            }
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!end || w == 0 || h == 0) {
        vrm_set_error("%s: truncated PNG", path);
        return VRM_ERR_IO;
    }
    *width = w;
    *height = h;
    *channels = ch;
    const size_t row = (size_t)w * ch, need = row * h;
    // a deflate stream expands at most ~1032:1: a header that promises more than the IDAT data can hold is corrupt, and
    // must not size an allocation
    if ((row + 1) * (size_t)h > idat.size() * 1040 + 1024) {
        vrm_set_error("%s: header says %u x %u x %u but the file holds only %zu bytes of image data", path, w, h, ch, idat.size());
        return VRM_ERR_IO;
    }
    if (pixels == nullptr) return VRM_OK; // size query
    REQUIRE(capacity >= need, "%s needs %zu bytes, buffer holds %zu", path, need, capacity);
    std::vector<uint8_t> raw((row + 1) * h);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) {
        vrm_set_error("%s: zlib stream does not hold %zu bytes", path, raw.size());
        return VRM_ERR_IO;
    }
    // undo the per-row filters (PNG spec 9.2) in place into `pixels`
    const size_t bpp = ch;
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = raw[(row + 1) * y];
        const uint8_t *src = &raw[(row + 1) * y + 1];
        uint8_t *dst = pixels + row * y;
        const uint8_t *up = y ? dst - row : nullptr;
        for (size_t i = 0; i < row; ++i) {
            const int a = i >= bpp ? dst[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            switch (ft) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default:
                vrm_set_error("%s: bad filter type %d in row %u", path, ft, y);
                return VRM_ERR_IO;
            }
            dst[i] = (uint8_t)(src[i] + pred);
        }
    }
    return VRM_OK;
    });
}
