// vrm_api.cpp -- the C ABI of include/vrm.h: context, voxel grid, render targets, draw.
// Compiled with hipcc; uses the HIP runtime API only (no torch, no CPU fallback).
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include <dlfcn.h>
#include <math.h>

#include "vrm_internal.h"

static thread_local char g_err[512] = "";

void vrm_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int vrm_check_hip(hipError_t e, const char *what)
{
    if (e == hipSuccess) return VRM_OK;
    vrm_set_error("%s: %s", what, hipGetErrorString(e));
    return VRM_ERR_HIP;
}

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

static const uint32_t kViewRing = 4096; // descriptor ring slots

// ---- the allocation-free draw (include/vrm.h: vrm_draw_batch_static) ------------------------------------------------
// Scratch of the march kernel = the view descriptors of the batch (VrmViewDesc: inverse view-projection, position,
// output pointers), which vrm_draw_batch stages in a ring the library hipMalloc'ed at vrm_create.  Here they live in
// a file-scope device array: every GPU this library runs on holds its own copy (30 KiB) until the process ends.  The
// array's NAME is a host-side stand-in, not a device address; the address on each device comes from
// hipGetSymbolAddress, once per device with that device current (scratch_init, called by vrm_create), and only that
// pointer is ever handed to a copy or a kernel.
#define VRM_STATIC_MAX_VIEWS 256
static __device__ __align__(256) unsigned char scratch[VRM_STATIC_MAX_VIEWS * sizeof(VrmViewDesc)];
static const int kMaxDevices = 64;
static void *g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mu;

static int scratch_init(int device, void **out)
{
    REQUIRE(device >= 0 && device < kMaxDevices, "device ordinal %d beyond the %d this library tracks", device, kMaxDevices);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void *p = nullptr;
        HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch))); // the current device's copy (the caller made `device` current)
        REQUIRE(p != nullptr, "hipGetSymbolAddress returned a null address for the descriptor scratch");
        g_scratch_ptr[device] = p;
    }
    *out = g_scratch_ptr[device];
    return VRM_OK;
}

extern "C" const char *vrm_last_error(void) { return g_err; }
extern "C" const char *vrm_version(void) { return "vrm 0.1 (gfx950)"; }

// RaymarchTransferFunctionColors::new (grid.rs:282-290)
static void tf_defaults(vrm_transfer_function *tf)
{
    const float a[4] = {0.117f, 0.188f, 0.62f, 1.0f}, b[4] = {0.7294f, 0.7294f, 0.7294f, 1.0f}, c[4] = {0.5725f, 0.0f, 0.039f, 1.0f};
    memcpy(tf->color_a, a, 16);
    memcpy(tf->color_b, b, 16);
    memcpy(tf->color_c, c, 16);
    memset(tf->use_transfer_function, 0, 16);
}

static int bind(vrm_ctx *ctx)
{
    REQUIRE(ctx != nullptr, "null context");
    return vrm_check_hip(hipSetDevice(ctx->device), "hipSetDevice");
}

static void grid_free(vrm_ctx *ctx)
{
    if (ctx->vox) (void)hipFree(ctx->vox);
    if (ctx->occ) (void)hipFree(ctx->occ);
    if (ctx->bricks) (void)hipFree(ctx->bricks);
    if (ctx->abricks) (void)hipFree(ctx->abricks);
    ctx->vox = nullptr;
    ctx->occ = nullptr;
    ctx->bricks = nullptr;
    ctx->abricks = nullptr;
    ctx->vox_words = 0;
}

// ---- compact form of a scalar volume -------------------------------------------------------------
// A scalar volume (r == g == b == a in every voxel: DAT files, 1-channel f32) is marched from its packed alpha bricks and
// the occupancy pyramid alone -- 1 B/voxel -- so after an upload that turned out scalar the x-fastest RGBA8 copy and the RGBA
// bricks (8 B/voxel) are released.  They come back on demand (grid_expand: [b,b,b,b] from the alpha bricks, exact) when
// something needs them: a download, a broadcast, the voxel-space compare, or a draw with a kernel that stages RGBA texels.
static size_t rgba_bricks_bytes(const uint32_t d[3]) { return (size_t)((d[0] + 3) / 4) * ((d[1] + 3) / 4) * ((d[2] + 1) / 2) * 128; }

// storage for the 4 B/voxel copies (contents undefined): what every upload writes into
static int grid_rgba_storage(vrm_ctx *ctx)
{
    // all or nothing: a `vox` without `bricks` left behind by a failed second allocation would later pass for a valid copy
    const bool had_vox = ctx->vox != nullptr;
    if (!ctx->vox) HIP_TRY(hipMalloc((void **)&ctx->vox, ctx->vox_words * 4));
    if (!ctx->bricks) {
        const int rc = vrm_check_hip(hipMalloc((void **)&ctx->bricks, rgba_bricks_bytes(ctx->dim)), "hipMalloc");
        if (rc) {
            ctx->bricks = nullptr;
            if (!had_vox) {
                (void)hipFree(ctx->vox);
                ctx->vox = nullptr;
            }
            return rc;
        }
    }
    return VRM_OK;
}

static int grid_expand(vrm_ctx *ctx)
{
    if (ctx->vox && ctx->bricks && !ctx->lazy_default) return VRM_OK;
    REQUIRE(ctx->abricks != nullptr, "the context holds no voxel grid");
    const bool had_vox = ctx->vox != nullptr;
    int rc = grid_rgba_storage(ctx);
    if (rc) return rc;
    if (ctx->lazy_default) {
        // a grid that was resized and never filled: now somebody looks at it -- Voxel::default = [0,0,0,255] (voxel.rs:13-16)
        HIP_TRY(vrm_launch_fill(ctx->vox, ctx->vox_words, 0xFF000000u, ctx->stream));
        HIP_TRY(vrm_launch_build_occupancy(ctx->vox, 0, ctx->dim, ctx->occ, ctx->stream));
        HIP_TRY(vrm_launch_build_bricks(ctx->vox, ctx->dim, ctx->bricks, ctx->stream));
        HIP_TRY(vrm_launch_build_alpha_bricks(ctx->vox, 0, ctx->dim, ctx->abricks, ctx->stream));
        ctx->lazy_default = false;
        return VRM_OK;
    }
    if (!had_vox) HIP_TRY(vrm_launch_expand_alpha_bricks(ctx->abricks, ctx->dim, ctx->vox, ctx->stream));
    HIP_TRY(vrm_launch_build_bricks(ctx->vox, ctx->dim, ctx->bricks, ctx->stream));
    return VRM_OK;
}


// does the march, as the options stand, read anything but the packed alpha bricks and the pyramid?
static bool march_needs_rgba(const vrm_ctx *ctx, bool scalar)
{
    const bool one_channel = scalar || ctx->tf.use_transfer_function[0] != 0;
    return !(one_channel && ctx->opt_packed != 0 && (ctx->opt_kernel == 1 || ctx->opt_kernel == 3 || ctx->opt_kernel == 4));
}
static bool draw_needs_rgba(const vrm_ctx *ctx) { return march_needs_rgba(ctx, ctx->scalar); }

// what every consumer of the volume calls first: a grid that was resized and never filled gets its default contents now, and
// a compact scalar volume its RGBA copies if `need_rgba` (or the options of the march) ask for them
static int grid_ready(vrm_ctx *ctx, bool need_rgba)
{
    if (!ctx->abricks) {
        vrm_set_error("the context holds no voxel grid (an earlier vrm_grid_resize failed)");
        return VRM_ERR_STATE;
    }
    if (ctx->lazy_default || ((!ctx->vox || !ctx->bricks) && (need_rgba || draw_needs_rgba(ctx)))) return grid_expand(ctx);
    return VRM_OK;
}

int vrm_grid_expand_internal(vrm_ctx *ctx) { return grid_ready(ctx, true); }

static int grid_compact(vrm_ctx *ctx)
{
    if (!ctx->scalar || !ctx->opt_compact || draw_needs_rgba(ctx)) return VRM_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // the builders of the packed bricks and of the pyramid read the RGBA copy
    if (ctx->vox) (void)hipFree(ctx->vox);
    if (ctx->bricks) (void)hipFree(ctx->bricks);
    ctx->vox = nullptr;
    ctx->bricks = nullptr;
    return VRM_OK;
}

static int grid_alloc_buffers(vrm_ctx *ctx, uint32_t x, uint32_t y, uint32_t z)
{
    const size_t words = (size_t)x * y * z;
    const uint32_t xyz[3] = {x, y, z};
    // (the x-fastest RGBA8 copy and the RGBA bricks, 8 B/voxel, are allocated by whoever first needs them: grid_rgba_storage)
    uint32_t occ_off[VRM_OCC_LEVELS];
    HIP_TRY(hipMalloc((void **)&ctx->occ, vrm_occupancy_bytes(xyz, occ_off)));
    HIP_TRY(hipMalloc((void **)&ctx->abricks, vrm_alpha_bricks_bytes(xyz)));
    ctx->vox_words = words;
    ctx->dim[0] = x; ctx->dim[1] = y; ctx->dim[2] = z;
    ctx->gdim[0] = x; ctx->gdim[1] = y; ctx->gdim[2] = z;
    ctx->z_store0 = 0;
    ctx->slab_z0 = 0;
    ctx->slab_z1 = z;
    return VRM_OK;
}

static int grid_reset_contents(vrm_ctx *ctx);

// The old volume is released BEFORE the new one is allocated (a 64 GiB slab and its successor need not fit side by
// side).  If the new one cannot be allocated the context falls back to the start-up grid (3x3x3 default voxels,
// ray_marcher/mod.rs:29) so that it stays usable, and the call reports the allocation failure.
static int grid_alloc(vrm_ctx *ctx, uint32_t x, uint32_t y, uint32_t z)
{
    REQUIRE(x > 0 && y > 0 && z > 0, "grid dimensions must be non-zero (got %u x %u x %u)", x, y, z);
    REQUIRE((size_t)x * y * z < ((size_t)1 << 40), "grid too large");
    // the staging offsets inside one brick layer are 32-bit: 8 bytes per texel of a z slice must stay below 4 GiB
    REQUIRE((size_t)x * y <= ((size_t)1 << 28), "a z slice of %u x %u texels exceeds 2^28 (16384 x 16384)", x, y);
    // draws in flight still read the old volume: they are on ctx->stream, so order the free behind them
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    grid_free(ctx);
    int rc = grid_alloc_buffers(ctx, x, y, z);
    if (rc == VRM_OK) return rc;
    char why[512];
    snprintf(why, sizeof why, "%s", vrm_last_error());
    (void)hipGetLastError(); // clear the sticky allocation error
    grid_free(ctx);
    if (grid_alloc_buffers(ctx, 3, 3, 3) == VRM_OK && grid_reset_contents(ctx) == VRM_OK)
        vrm_set_error("cannot allocate a %u x %u x %u grid (%s); the context now holds the 3 x 3 x 3 start-up grid", x, y, z, why);
    else
        grid_free(ctx);
    return rc;
}

static int grid_reset_contents(vrm_ctx *ctx)
{
    // Voxel::default = [0,0,0,255] (voxel.rs:13-16) -- as a promise: VoxelGrid::new is always followed by a loader that
    // overwrites every voxel (dat.rs:21, netcdf.rs:47), so the fill and its 8 B/voxel of RGBA storage wait until something
    // looks at the grid before an upload has replaced it (grid_expand); a scalar upload then never holds the RGBA form at all
    ctx->lazy_default = true;
    ctx->scalar = false;
    ctx->attenuation = 1.0f;   // grid.rs:100,197
    tf_defaults(&ctx->tf);     // grid.rs:198
    return VRM_OK;
}

extern "C" int vrm_grid_resize(vrm_ctx *ctx, uint32_t x, uint32_t y, uint32_t z)
{
    int rc = bind(ctx);
    if (rc) return rc;
    rc = grid_alloc(ctx, x, y, z);
    if (rc) return rc;
    return grid_reset_contents(ctx);
}

extern "C" int vrm_grid_resize_slab(vrm_ctx *ctx, uint32_t x, uint32_t y, uint32_t z, uint32_t z_begin, uint32_t z_end)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(x > 0 && y > 0 && z > 0, "grid dimensions must be non-zero (got %u x %u x %u)", x, y, z);
    REQUIRE(z_begin < z_end && z_end <= z, "slab [%u, %u) is not a non-empty range inside [0, %u)", z_begin, z_end, z);
    // one ghost slice either side: a sample of slice range [z_begin, z_end) filters texels z_begin-1 .. z_end
    const uint32_t s0 = z_begin > 0 ? z_begin - 1 : 0, s1 = z_end < z ? z_end + 1 : z;
    rc = grid_alloc(ctx, x, y, s1 - s0);
    if (rc) return rc;
    ctx->gdim[2] = z;
    ctx->z_store0 = s0;
    ctx->slab_z0 = z_begin;
    ctx->slab_z1 = z_end;
    return grid_reset_contents(ctx);
}

extern "C" int vrm_grid_resident_bytes(vrm_ctx *ctx, size_t *bytes)
{
    REQUIRE(ctx != nullptr && bytes != nullptr, "null argument");
    uint32_t off[VRM_OCC_LEVELS];
    size_t n = 0;
    if (ctx->vox) n += ctx->vox_words * 4;
    if (ctx->bricks) n += rgba_bricks_bytes(ctx->dim);
    if (ctx->abricks) n += vrm_alpha_bricks_bytes(ctx->dim);
    if (ctx->occ) n += vrm_occupancy_bytes(ctx->dim, off);
    *bytes = n;
    return VRM_OK;
}

extern "C" int vrm_grid_get_slab(vrm_ctx *ctx, uint32_t *z_store_begin, uint32_t *z_store_end, uint32_t *z_begin, uint32_t *z_end)
{
    REQUIRE(ctx != nullptr, "null context");
    if (z_store_begin) *z_store_begin = ctx->z_store0;
    if (z_store_end) *z_store_end = ctx->z_store0 + ctx->dim[2];
    if (z_begin) *z_begin = ctx->slab_z0;
    if (z_end) *z_end = ctx->slab_z1;
    return VRM_OK;
}

extern "C" int vrm_create(const vrm_config *config, vrm_ctx **out_ctx)
{
    REQUIRE(config != nullptr && out_ctx != nullptr, "vrm_create: null argument");
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        vrm_set_error("no HIP device available (%s); this library has no CPU fallback",
                      e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return VRM_ERR_NO_DEVICE;
    }
    REQUIRE(config->device_id >= 0 && config->device_id < n, "device_id %d out of range (have %d)", config->device_id, n);
    vrm_ctx *ctx = new (std::nothrow) vrm_ctx();
    REQUIRE(ctx != nullptr, "out of host memory");
    ctx->device = config->device_id;
    ctx->def_width = config->width ? config->width : 1024;   // lib.rs:34 default window
    ctx->def_height = config->height ? config->height : 1024;
    tf_defaults(&ctx->tf);
    int rc = VRM_OK;
    do {
        if ((rc = vrm_check_hip(hipSetDevice(ctx->device), "hipSetDevice"))) break;
        if ((rc = vrm_check_hip(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking), "hipStreamCreate"))) break;
        if ((rc = vrm_check_hip(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking), "hipStreamCreate"))) break;
        if ((rc = scratch_init(ctx->device, &ctx->scratch))) break; // this device's copy of the static descriptor scratch
        if ((rc = vrm_check_hip(hipMalloc((void **)&ctx->flag_dev, 256), "hipMalloc"))) break;
        if ((rc = vrm_check_hip(hipMemsetAsync(ctx->flag_dev, 0, 256, ctx->stream), "hipMemsetAsync"))) break;
        if ((rc = vrm_check_hip(hipMalloc((void **)&ctx->queue_dev, VRM_QUEUE_BYTES), "hipMalloc"))) break;
        for (uint32_t i = 0; i < vrm_ctx::kBatchEvents && !rc; ++i)
            rc = vrm_check_hip(hipEventCreateWithFlags(&ctx->batch_events[i], hipEventDisableTiming), "hipEventCreate");
        if (rc) break;
        if ((rc = vrm_check_hip(hipMalloc((void **)&ctx->views_dev, sizeof(VrmViewDesc) * kViewRing), "hipMalloc"))) break;
        if ((rc = vrm_check_hip(hipHostMalloc((void **)&ctx->views_host, sizeof(VrmViewDesc) * kViewRing, hipHostMallocDefault),
                                "hipHostMalloc")))
            break;
        ctx->views_cap = 0; // ring write position
        if ((rc = vrm_grid_resize(ctx, 3, 3, 3))) break; // ray_marcher/mod.rs:29
        if ((rc = grid_expand(ctx))) break; // (the start-up grid is materialised at once: a fresh context can be drawn by the entry point that may not allocate)
    } while (0);
    if (rc) {
        vrm_destroy(ctx);
        return rc;
    }
    *out_ctx = ctx;
    return VRM_OK;
}

extern "C" void vrm_destroy(vrm_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    // render targets that outlive the context (the reference's Screenshotter is a separate object too, state.rs:161):
    // nothing of theirs is in flight any more; they keep their memory until vrm_image_destroy and refuse every other call
    for (vrm_image *img : ctx->images) img->ctx = nullptr;
    ctx->images.clear();
    (void)vrm_comm_destroy(ctx);
    for (auto &p : ctx->prof_events) {
        (void)hipEventDestroy(p.first);
        (void)hipEventDestroy(p.second);
    }
    for (hipEvent_t e : ctx->batch_events)
        if (e) (void)hipEventDestroy(e);
    if (ctx->vox) (void)hipFree(ctx->vox);
    if (ctx->occ) (void)hipFree(ctx->occ);
    if (ctx->bricks) (void)hipFree(ctx->bricks);
    if (ctx->abricks) (void)hipFree(ctx->abricks);
    if (ctx->flag_dev) (void)hipFree(ctx->flag_dev);
    if (ctx->queue_dev) (void)hipFree(ctx->queue_dev);
    if (ctx->views_dev) (void)hipFree(ctx->views_dev);
    if (ctx->views_host) (void)hipHostFree(ctx->views_host);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    delete ctx;
}

// after new voxel data: is the volume scalar (r == g == b == a)?  (selects the 1-channel filter)
static int grid_finish_upload(vrm_ctx *ctx)
{
    ctx->lazy_default = false; // (every voxel has just been written)
    HIP_TRY(vrm_launch_build_occupancy(ctx->vox, 0, ctx->dim, ctx->occ, ctx->stream));
    HIP_TRY(vrm_launch_build_bricks(ctx->vox, ctx->dim, ctx->bricks, ctx->stream));
    HIP_TRY(vrm_launch_build_alpha_bricks(ctx->vox, 0, ctx->dim, ctx->abricks, ctx->stream));
    HIP_TRY(hipMemsetAsync(ctx->flag_dev, 0, 4, ctx->stream));
    HIP_TRY(vrm_launch_detect_nonscalar(ctx->vox, ctx->vox_words, ctx->flag_dev, ctx->stream));
    uint32_t flag = 1;
    HIP_TRY(hipMemcpyAsync(&flag, ctx->flag_dev, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->scalar = (flag == 0);
    return grid_compact(ctx);
}

extern "C" int vrm_grid_upload_rgba8(vrm_ctx *ctx, const uint8_t *host_voxels, size_t nbytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(host_voxels != nullptr, "null voxel pointer");
    REQUIRE(nbytes == ctx->vox_words * 4, "expected %zu bytes (4*X*Y*Z), got %zu", ctx->vox_words * 4, nbytes);
    if (int rc_ = grid_rgba_storage(ctx)) return rc_; // (a compact scalar volume released it)
    HIP_TRY(hipMemcpyAsync(ctx->vox, host_voxels, nbytes, hipMemcpyHostToDevice, ctx->stream));
    return grid_finish_upload(ctx);
}

extern "C" int vrm_grid_upload_rgba8_device(vrm_ctx *ctx, const void *device_voxels, size_t nbytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(device_voxels != nullptr, "null voxel pointer");
    REQUIRE(nbytes == ctx->vox_words * 4, "expected %zu bytes (4*X*Y*Z), got %zu", ctx->vox_words * 4, nbytes);
    if (int rc_ = grid_rgba_storage(ctx)) return rc_; // (a compact scalar volume released it)
    HIP_TRY(hipMemcpyAsync(ctx->vox, device_voxels, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    return grid_finish_upload(ctx);
}

static int upload_f32_common(vrm_ctx *ctx, const float *dev_src, uint32_t channels)
{
    if (int rc_ = grid_rgba_storage(ctx)) return rc_; // (a compact scalar volume released it)
    HIP_TRY(vrm_launch_quantize_f32(dev_src, channels, ctx->vox_words, ctx->vox, ctx->stream));
    return grid_finish_upload(ctx);
}

extern "C" int vrm_grid_upload_f32(vrm_ctx *ctx, const float *host_czyx, uint32_t channels)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(host_czyx != nullptr, "null data pointer");
    REQUIRE(channels == 1 || channels == 4, "channels must be 1 or 4 (got %u)", channels);
    const size_t bytes = ctx->vox_words * channels * sizeof(float);
    float *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&tmp, bytes));
    rc = vrm_check_hip(hipMemcpyAsync(tmp, host_czyx, bytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    if (!rc) rc = upload_f32_common(ctx, tmp, channels);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    return rc;
}

extern "C" int vrm_grid_upload_f32_device(vrm_ctx *ctx, const void *device_czyx, uint32_t channels)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(device_czyx != nullptr, "null data pointer");
    REQUIRE(channels == 1 || channels == 4, "channels must be 1 or 4 (got %u)", channels);
    return upload_f32_common(ctx, (const float *)device_czyx, channels);
}

extern "C" int vrm_grid_upload_scalar(vrm_ctx *ctx, const uint8_t *host_raw, size_t nbytes, uint32_t bytes_per_sample)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(host_raw != nullptr, "null data pointer");
    REQUIRE(bytes_per_sample == 1 || bytes_per_sample == 2, "bytes_per_sample must be 1 (UCHAR) or 2 (USHORT)");
    REQUIRE(nbytes == ctx->vox_words * bytes_per_sample, "expected %zu bytes, got %zu", ctx->vox_words * bytes_per_sample, nbytes);
    uint8_t *tmp = nullptr;
    // Direct ingest: a DAT volume is scalar by construction (dat.rs:74 writes the one value to all four channels), and
    // with the default options the march reads only its packed alpha bricks and the pyramid -- so both are built straight
    // from the raw samples and the 8 B/voxel of RGBA copies are never allocated: the ingest peaks at the raw samples plus
    // 1 B/voxel instead of 9 B/voxel (what lets a 2048 x 2048 x 4096 volume in at 35 GB instead of 155).  Same bytes as
    // the long way round (tests/test_gpu_api.py).
    const bool direct = ctx->opt_compact != 0 && !march_needs_rgba(ctx, true);
    if (direct) {
        HIP_TRY(hipStreamSynchronize(ctx->stream)); // draws in flight may still read the copies released here
        if (ctx->vox) (void)hipFree(ctx->vox);
        if (ctx->bricks) (void)hipFree(ctx->bricks);
        ctx->vox = nullptr;
        ctx->bricks = nullptr;
        // (from here until the builders below have run the grid counts as never filled: should the staging buffer not fit, the
        // context is left with default voxels, not with the packed bricks of a volume whose colours were just released)
        ctx->lazy_default = true;
        ctx->scalar = false;
    }
    HIP_TRY(hipMalloc((void **)&tmp, nbytes));
    rc = vrm_check_hip(hipMemcpyAsync(tmp, host_raw, nbytes, hipMemcpyHostToDevice, ctx->stream), "hipMemcpyAsync");
    if (direct) {
        if (!rc) rc = vrm_check_hip(vrm_launch_build_occupancy(tmp, bytes_per_sample, ctx->dim, ctx->occ, ctx->stream), "occupancy");
        if (!rc) rc = vrm_check_hip(vrm_launch_build_alpha_bricks(tmp, bytes_per_sample, ctx->dim, ctx->abricks, ctx->stream), "alpha bricks");
        if (!rc) {
            ctx->lazy_default = false;
            ctx->scalar = true;
        }
    } else {
        if (!rc) rc = grid_rgba_storage(ctx);
        if (!rc) rc = vrm_check_hip(vrm_launch_quantize_scalar(tmp, bytes_per_sample, ctx->vox_words, ctx->vox, ctx->stream), "quantize");
        if (!rc) rc = grid_finish_upload(ctx);
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    return rc;
}

extern "C" int vrm_grid_download_rgba8(vrm_ctx *ctx, uint8_t *host_voxels, size_t nbytes)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(host_voxels != nullptr, "null voxel pointer");
    REQUIRE(nbytes == ctx->vox_words * 4, "expected %zu bytes (4*X*Y*Z), got %zu", ctx->vox_words * 4, nbytes);
    if ((rc = grid_ready(ctx, true))) return rc; // a compact scalar volume: [b,b,b,b] back from the packed alpha bricks
    HIP_TRY(hipMemcpyAsync(host_voxels, ctx->vox, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VRM_OK;
}

extern "C" int vrm_set_attenuation(vrm_ctx *ctx, float attenuation)
{
    REQUIRE(ctx != nullptr, "null context");
    ctx->attenuation = attenuation;
    return VRM_OK;
}

extern "C" int vrm_set_transfer_function(vrm_ctx *ctx, const vrm_transfer_function *tf)
{
    REQUIRE(ctx != nullptr && tf != nullptr, "null argument");
    ctx->tf = *tf;
    return VRM_OK;
}

// VoxelGridUniform::new (grid.rs:265-280), f32 arithmetic as written there
static void grid_uniform(const vrm_ctx *ctx, vrm_grid_uniform *u)
{
    // (of the whole volume: a slab-partitioned context marches the same box as every other rank)
    uint32_t mn = ctx->gdim[0] < ctx->gdim[1] ? ctx->gdim[0] : ctx->gdim[1];
    if (ctx->gdim[2] < mn) mn = ctx->gdim[2];
    for (int i = 0; i < 3; ++i) {
        u->dimensions[i] = ctx->gdim[i];
        u->box_size[i] = (float)ctx->gdim[i] / (float)mn;
        u->box_min[i] = -u->box_size[i] / 2.0f;
    }
    u->dimensions[3] = ctx->gdim[0];
    u->box_size[3] = 0.0f;
    u->box_min[3] = 0.0f;
    for (int i = 0; i < 4; ++i) u->buffer[i] = ctx->attenuation;
}

extern "C" int vrm_get_grid_uniform(vrm_ctx *ctx, vrm_grid_uniform *out)
{
    REQUIRE(ctx != nullptr && out != nullptr, "null argument");
    grid_uniform(ctx, out);
    return VRM_OK;
}

// ---- images -------------------------------------------------------------------------------------

extern "C" int vrm_image_create(vrm_ctx *ctx, uint32_t width, uint32_t height, vrm_image **out_image)
{
    return vrm_guard("vrm_image_create", VRM_ERR_INVALID, [&]() -> int {
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(out_image != nullptr, "null argument");
    *out_image = nullptr;
    REQUIRE(width > 0 && height > 0 && width <= 65536 && height <= 65536, "bad image size %u x %u", width, height);
    vrm_image *img = new (std::nothrow) vrm_image();
    REQUIRE(img != nullptr, "out of host memory");
    img->ctx = ctx;
    img->device = ctx->device;
    img->width = width;
    img->height = height;
    img->pitch = (((size_t)width * 4 + 255) / 256) * 256; // screenshot.rs:94-96
    do {
        if ((rc = vrm_check_hip(hipMalloc((void **)&img->dev, img->pitch * height), "hipMalloc"))) break;
        if ((rc = vrm_check_hip(hipHostMalloc((void **)&img->host, img->pitch * height, hipHostMallocDefault), "hipHostMalloc"))) break;
        if ((rc = vrm_check_hip(hipEventCreateWithFlags(&img->drawn, hipEventDisableTiming), "hipEventCreate"))) break;
        if ((rc = vrm_check_hip(hipEventCreateWithFlags(&img->copied, hipEventDisableTiming), "hipEventCreate"))) break;
        // a never-drawn target reads back as the transparent clear colour (mod.rs:124)
        if ((rc = vrm_check_hip(hipMemsetAsync(img->dev, 0, img->pitch * height, ctx->stream), "hipMemsetAsync"))) break;
        if ((rc = vrm_check_hip(hipEventRecord(img->drawn, ctx->stream), "hipEventRecord"))) break;
    } while (0);
    if (!rc) {
        try {
            ctx->images.push_back(img);
        } catch (const std::bad_alloc &) {
            vrm_set_error("out of host memory");
            rc = VRM_ERR_INVALID;
        }
    }
    if (rc) {
        vrm_image_destroy(img);
        return rc;
    }
    *out_image = img;
    return VRM_OK;
    });
}

extern "C" void vrm_image_destroy(vrm_image *img)
{
    if (!img) return;
    (void)hipSetDevice(img->device);
    if (img->ctx) { // (an orphan -- its context was destroyed first -- has nothing in flight: vrm_destroy drained the streams)
        (void)hipStreamSynchronize(img->ctx->stream);
        (void)hipStreamSynchronize(img->ctx->copy_stream);
        auto &v = img->ctx->images;
        v.erase(std::remove(v.begin(), v.end(), img), v.end());
    }
    if (img->dev) (void)hipFree(img->dev);
    if (img->host) (void)hipHostFree(img->host);
    if (img->diag_rgba) (void)hipFree(img->diag_rgba);
    if (img->diag_iters) (void)hipFree(img->diag_iters);
    if (img->diag_geom) (void)hipFree(img->diag_geom);
    if (img->drawn) (void)hipEventDestroy(img->drawn);
    if (img->copied) (void)hipEventDestroy(img->copied);
    delete img;
}

#define REQUIRE_LIVE(img)                                                                              \
    do {                                                                                               \
        if ((img)->ctx == nullptr) {                                                                   \
            vrm_set_error("the image's context was destroyed; only vrm_image_destroy is valid on it"); \
            return VRM_ERR_STATE;                                                                      \
        }                                                                                              \
    } while (0)

extern "C" int vrm_image_readback_async(vrm_image *img)
{
    REQUIRE(img != nullptr, "null image");
    REQUIRE_LIVE(img);
    int rc = bind(img->ctx);
    if (rc) return rc;
    if (img->mapped) {
        vrm_set_error("image is mapped; unmap before the next readback");
        return VRM_ERR_STATE;
    }
    vrm_ctx *ctx = img->ctx;
    HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, img->drawn_batch ? img->drawn_batch : img->drawn, 0));
    HIP_TRY(hipMemcpyAsync(img->host, img->dev, img->pitch * img->height, hipMemcpyDeviceToHost, ctx->copy_stream));
    HIP_TRY(hipEventRecord(img->copied, ctx->copy_stream));
    img->copy_pending = true;
    img->copy_valid = false;
    return VRM_OK;
}

extern "C" int vrm_image_map(vrm_image *img, const uint8_t **rgba8, size_t *pitch)
{
    REQUIRE(img != nullptr && rgba8 != nullptr && pitch != nullptr, "null argument");
    REQUIRE_LIVE(img);
    int rc = bind(img->ctx);
    if (rc) return rc;
    if (img->mapped) {
        vrm_set_error("image already mapped");
        return VRM_ERR_STATE;
    }
    if (!img->copy_pending && !img->copy_valid) {
        rc = vrm_image_readback_async(img);
        if (rc) return rc;
    }
    if (img->copy_pending) {
        HIP_TRY(hipEventSynchronize(img->copied));
        img->copy_pending = false;
        img->copy_valid = true;
    }
    img->mapped = true;
    *rgba8 = img->host;
    *pitch = img->pitch;
    return VRM_OK;
}

extern "C" int vrm_image_unmap(vrm_image *img)
{
    REQUIRE(img != nullptr, "null image");
    if (!img->mapped) {
        vrm_set_error("image is not mapped");
        return VRM_ERR_STATE;
    }
    img->mapped = false;
    return VRM_OK;
}

extern "C" int vrm_image_save_png(vrm_image *img, const char *path)
{
    REQUIRE(img != nullptr && path != nullptr, "null argument");
    const uint8_t *px = nullptr;
    size_t pitch = 0;
    int rc = vrm_image_map(img, &px, &pitch);
    if (rc) return rc;
    rc = vrm_write_png_rgba8(path, px, img->width, img->height, pitch);
    (void)vrm_image_unmap(img);
    return rc;
}

extern "C" int vrm_image_read_ray_diagnostics(vrm_image *img, float *geom8)
{
    REQUIRE(img != nullptr && geom8 != nullptr, "null argument");
    REQUIRE_LIVE(img);
    int rc = bind(img->ctx);
    if (rc) return rc;
    if (!img->diag_geom) {
        vrm_set_error("no diagnostics: set VRM_OPT_DIAGNOSTICS before drawing into this image");
        return VRM_ERR_STATE;
    }
    HIP_TRY(hipMemcpyAsync(geom8, img->diag_geom, (size_t)img->width * img->height * 32, hipMemcpyDeviceToHost, img->ctx->stream));
    HIP_TRY(hipStreamSynchronize(img->ctx->stream));
    return VRM_OK;
}

extern "C" int vrm_image_read_diagnostics(vrm_image *img, float *rgba_f32, uint32_t *iterations)
{
    REQUIRE(img != nullptr, "null image");
    REQUIRE_LIVE(img);
    int rc = bind(img->ctx);
    if (rc) return rc;
    if (!img->diag_rgba || !img->diag_iters) {
        vrm_set_error("no diagnostics: set VRM_OPT_DIAGNOSTICS before drawing into this image");
        return VRM_ERR_STATE;
    }
    vrm_ctx *ctx = img->ctx;
    const size_t n = (size_t)img->width * img->height;
    if (rgba_f32) HIP_TRY(hipMemcpyAsync(rgba_f32, img->diag_rgba, n * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (iterations) HIP_TRY(hipMemcpyAsync(iterations, img->diag_iters, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (iterations)
        for (size_t i = 0; i < n; ++i) iterations[i] &= 0xFFFFu; // the high half holds the executed-sample count
    return VRM_OK;
}

extern "C" int vrm_image_read_sample_counts(vrm_image *img, uint32_t *executed)
{
    REQUIRE(img != nullptr && executed != nullptr, "null argument");
    REQUIRE_LIVE(img);
    int rc = bind(img->ctx);
    if (rc) return rc;
    if (!img->diag_iters) {
        vrm_set_error("no diagnostics: set VRM_OPT_DIAGNOSTICS before drawing into this image");
        return VRM_ERR_STATE;
    }
    vrm_ctx *ctx = img->ctx;
    const size_t n = (size_t)img->width * img->height;
    HIP_TRY(hipMemcpyAsync(executed, img->diag_iters, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < n; ++i) executed[i] >>= 16;
    return VRM_OK;
}

// ---- draw -----------------------------------------------------------------------------------------

static void fill_params(const vrm_ctx *ctx, VrmMarchParams *p, uint32_t w, uint32_t h, size_t pitch)
{
    vrm_grid_uniform u;
    grid_uniform(ctx, &u);
    VrmGridDev &g = p->grid;
    g.vox = ctx->vox;
    g.scalar = ctx->scalar ? 1u : 0u;
    g.occ = ctx->occ;
    (void)vrm_occupancy_bytes(ctx->dim, g.occ_off);
    g.bricks = ctx->bricks;
    g.nbx = (ctx->dim[0] + 3) / 4;
    g.abricks = ctx->abricks;
    g.nbxa = (ctx->dim[0] + 15) / 16;
    g.nby = (ctx->dim[1] + 3) / 4;
    for (int i = 0; i < 3; ++i) {
        g.dim[i] = ctx->dim[i];
        g.occ_dim[i] = (ctx->dim[i] + 3) / 4;
        g.box_min[i] = u.box_min[i];
        g.box_size[i] = u.box_size[i];
        g.box_max[i] = u.box_min[i] + u.box_size[i]; // wgsl:202,250
        g.tex_scale[i] = (float)ctx->gdim[i] / u.box_size[i];
        g.tex_bias[i] = -u.box_min[i] * g.tex_scale[i] - 0.5f;
        g.u_lo[i] = -0.5f;
        g.u_hi[i] = (float)ctx->dim[i] - 0.5f;
    }
    // slab partition: texel z relative to the first stored slice; membership bounds in world space, the same
    // expression on every rank so that neighbouring slabs share one boundary value and each sample has one owner
    const float Zf = (float)ctx->gdim[2];
    g.slab = (ctx->slab_z0 != 0 || ctx->slab_z1 != ctx->gdim[2]) ? 1u : 0u;
    g.tex_bias[2] -= (float)ctx->z_store0;
    g.u_lo[2] = (float)ctx->slab_z0 - 0.5f - (float)ctx->z_store0;
    g.u_hi[2] = (float)ctx->slab_z1 - 0.5f - (float)ctx->z_store0;
    g.in_min_z = ctx->slab_z0 == 0 ? g.box_min[2] : u.box_min[2] + ((float)ctx->slab_z0 / Zf) * u.box_size[2];
    g.in_max_z = ctx->slab_z1 == ctx->gdim[2] ? g.box_max[2] : u.box_min[2] + ((float)ctx->slab_z1 / Zf) * u.box_size[2];
    p->width = w;
    p->height = h;
    p->pitch_px = (uint32_t)(pitch / 4);
    p->tiles_x = (w + 15) / 16;
    p->tiles_y = (h + 15) / 16;
    p->sblocks_x = (((w + 63) / 64) + 7u) & ~7u; // 64x64-pixel super-blocks, columns padded to a multiple of 8 (XCD diagonals)
    p->sblocks_y = (h + 63) / 64;
    p->attenuation = u.buffer[0];
    for (int i = 0; i < 3; ++i) {
        p->col_a[i] = ctx->tf.color_a[i];
        p->col_b[i] = ctx->tf.color_b[i];
        p->col_c[i] = ctx->tf.color_c[i];
    }
    p->use_tf = ctx->tf.use_transfer_function[0] != 0 ? 1u : 0u;
    p->chunk_steps = (uint32_t)ctx->opt_chunk;
    p->packed_alpha = (int32_t)ctx->opt_packed;
    p->queue = ctx->opt_kernel == 4 ? nullptr : ctx->queue_dev; // (4: one workgroup per tile whatever the launcher would choose)
}

// does the batch hold a view the FAR instance of the march kernel must take?  (The kernels decide per view at 4096 texels;
// a lower bar here only means a launch whose waves all leave at once.)
static uint32_t any_far_view(const VrmMarchParams &p, const vrm_camera_uniform *cameras, uint32_t n)
{
    for (uint32_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a)
            if (!(fabsf(fmaf(cameras[i].position[a], p.grid.tex_scale[a], p.grid.tex_bias[a])) <= 4000.0f)) return 1u;
    return 0u;
}

// rays per lane of the packed scalar march (VRM_OPT_RAYS_PER_LANE): two for volumes larger than the Infinity Cache where, in every
// view of the batch, a pixel is narrower than 0.6 texel at the volume's centre -- the width of a pixel taken from the directions of two x-adjacent rays at the image
// centre, by fs_main's own formula (wgsl:98-108).  The launcher drops to one for far or slab-partitioned views.
static uint32_t rays_for_batch(const vrm_ctx *ctx, const VrmMarchParams &p, const vrm_camera_uniform *cameras, uint32_t n)
{
    if (ctx->opt_rays == 1 || ctx->opt_rays == 2) return (uint32_t)ctx->opt_rays;
    // Measured (profiles/r3/rays_per_lane.txt): -9.5 % for a 1024^3 volume at 4096^2, nothing for 512^3 and 256^3 at 2048^2
    // (what two rays save is staging traffic and prologues; a volume that sits in the 256 MB Infinity Cache is bound by its
    // step loops), and +13 % at one texel per pixel.  So: large volumes only.
    if (p.width < 16 || (uint64_t)p.grid.dim[0] * p.grid.dim[1] * p.grid.dim[2] <= (256ull << 20)) return 1u;
    float scale = 0.0f;
    for (int a = 0; a < 3; ++a) scale = fmaxf(scale, p.grid.tex_scale[a]);
    for (uint32_t i = 0; i < n; ++i) {
        const float *m = cameras[i].inverse_view_proj;
        float d[2][3];
        for (int k = 0; k < 2; ++k) {
            const float sx = (((float)(p.width / 2 + (uint32_t)k) + 0.5f) / (float)p.width) * 2.0f - 1.0f;
            const float sy = (1.0f - ((float)(p.height / 2) + 0.5f) / (float)p.height) * 2.0f - 1.0f;
            float len = 0.0f;
            for (int c = 0; c < 3; ++c) {
                d[k][c] = m[0 + c] * sx + m[4 + c] * sy + m[8 + c] + m[12 + c];
                len += d[k][c] * d[k][c];
            }
            len = sqrtf(len);
            for (int c = 0; c < 3; ++c) d[k][c] /= len;
        }
        float angle = 0.0f, dist = 0.0f;
        for (int c = 0; c < 3; ++c) {
            angle += (d[1][c] - d[0][c]) * (d[1][c] - d[0][c]);
            const float centre = p.grid.box_min[c] + 0.5f * p.grid.box_size[c];
            dist += (cameras[i].position[c] - centre) * (cameras[i].position[c] - centre);
        }
        const float texels_per_pixel = sqrtf(angle) * sqrtf(dist) * scale;
        if (!(texels_per_pixel <= 0.6f)) return 1u; // (NaN: one)
    }
    return 2u;
}

static int draw_batch_impl(vrm_ctx *ctx, const vrm_camera_uniform *cameras, uint32_t n, vrm_image *const *targets, bool use_static)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(cameras != nullptr && targets != nullptr, "null argument");
    REQUIRE(n > 0 && n <= kViewRing && n <= 65535, "batch size %u out of range (1..%u)", n, kViewRing);
    if (use_static) {
        REQUIRE(n <= VRM_STATIC_MAX_VIEWS, "vrm_draw_batch_static: %u views exceed the static descriptor scratch (at most %d per call)",
                n, VRM_STATIC_MAX_VIEWS);
        REQUIRE(ctx->scratch != nullptr, "the descriptor scratch of device %d was not initialised", ctx->device);
    }
    if (!ctx->abricks) {
        vrm_set_error("the context holds no voxel grid (an earlier vrm_grid_resize failed)");
        return VRM_ERR_STATE;
    }
    if (ctx->lazy_default || ((!ctx->vox || !ctx->bricks) && draw_needs_rgba(ctx))) {
        // a grid nobody has filled yet, or a compact scalar volume and options that stage RGBA texels (VRM_OPT_KERNEL 0 / 2,
        // VRM_OPT_PACKED_ALPHA 0): materialise / expand it
        if (use_static) {
            vrm_set_error("vrm_draw_batch_static: the volume is held in a compact form (a scalar volume, or a grid that was resized and "
                          "never filled) and this draw needs its RGBA copies, which this entry point may not allocate (draw once with "
                          "vrm_draw_batch, or keep the default kernel options)");
            return VRM_ERR_STATE;
        }
        if ((rc = grid_expand(ctx))) return rc;
    }
    const uint32_t w = targets[0] ? targets[0]->width : 0, h = targets[0] ? targets[0]->height : 0;
    for (uint32_t i = 0; i < n; ++i) {
        REQUIRE(targets[i] != nullptr && targets[i]->ctx == ctx, "target %u is null or belongs to another context", i);
        REQUIRE(targets[i]->width == w && targets[i]->height == h, "all targets of a batch must have the same size");
        if (targets[i]->mapped) {
            vrm_set_error("target %u is mapped", i);
            return VRM_ERR_STATE;
        }
    }
    // A target whose read-back is still in flight on the copy stream must not be overwritten under it: the draw waits
    // for the copy (the reference's buffer is mapped/unmapped around each copy, screenshot.rs:58-91, so the case cannot
    // arise there).  The copy itself stays valid: it holds the image as it was when the read-back was enqueued.
    for (uint32_t i = 0; i < n; ++i)
        if (targets[i]->copy_pending) HIP_TRY(hipStreamWaitEvent(ctx->stream, targets[i]->copied, 0));
    // (the allocation-free form neither allocates diagnostics buffers nor creates profiling events: both switches are
    // ignored there)
    const bool diag = ctx->opt_diag != 0 && !use_static;
    if (diag) {
        for (uint32_t i = 0; i < n; ++i) {
            vrm_image *img = targets[i];
            if (!img->diag_rgba) HIP_TRY(hipMalloc((void **)&img->diag_rgba, (size_t)w * h * 16));
            if (!img->diag_iters) HIP_TRY(hipMalloc((void **)&img->diag_iters, (size_t)w * h * 4));
            if (!img->diag_geom) HIP_TRY(hipMalloc((void **)&img->diag_geom, (size_t)w * h * 32));
        }
    }
    // descriptor ring: wrap => everything that used the old slots must have finished
    if (ctx->views_cap + n > kViewRing) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->views_cap = 0;
    }
    const uint32_t slot = ctx->views_cap;
    ctx->views_cap += n;
    for (uint32_t i = 0; i < n; ++i) {
        VrmViewDesc &d = ctx->views_host[slot + i];
        memcpy(d.ivp, cameras[i].inverse_view_proj, 64);
        memcpy(d.pos, cameras[i].position, 16);
        d.out = targets[i]->dev;
        d.diag_rgba = diag ? targets[i]->diag_rgba : nullptr;
        d.diag_iters = diag ? targets[i]->diag_iters : nullptr;
        d.diag_geom = diag ? targets[i]->diag_geom : nullptr;
        d.partial_rgba = nullptr;
    }
    // the pinned source slots advance through the ring either way (an async copy reads them later); the device side is
    // the ring slot, or always the start of the static scratch -- stream order keeps the previous launch's reads ahead
    // of this copy
    VrmViewDesc *dev_views = use_static ? reinterpret_cast<VrmViewDesc *>(ctx->scratch) : ctx->views_dev + slot;
    HIP_TRY(hipMemcpyAsync(dev_views, ctx->views_host + slot, sizeof(VrmViewDesc) * n, hipMemcpyHostToDevice, ctx->stream));

    VrmMarchParams p;
    memset(&p, 0, sizeof p);
    fill_params(ctx, &p, w, h, targets[0]->pitch);
    p.any_far = any_far_view(p, cameras, n);
    p.rays = rays_for_batch(ctx, p, cameras, n);
    p.views = dev_views;
    p.diag_counters = diag ? reinterpret_cast<unsigned long long *>(ctx->flag_dev + 16) : nullptr; // bytes 64..127 of the 256-byte scratch

    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->opt_profile && !use_static) {
        if (ctx->prof_used == ctx->prof_events.size()) {
            hipEvent_t a, b;
            HIP_TRY(hipEventCreate(&a));
            HIP_TRY(hipEventCreate(&b));
            ctx->prof_events.emplace_back(a, b);
        }
        e0 = ctx->prof_events[ctx->prof_used].first;
        e1 = ctx->prof_events[ctx->prof_used].second;
        ctx->prof_used++;
        HIP_TRY(hipEventRecord(e0, ctx->stream));
    }
    HIP_TRY(vrm_launch_march(p, n, (int)ctx->opt_kernel, diag, ctx->opt_skip != 0, ctx->stream));
    if (e1) HIP_TRY(hipEventRecord(e1, ctx->stream));
    // one event behind the launch for all its targets (see vrm_image::drawn_batch)
    hipEvent_t ev = ctx->batch_events[ctx->batch_next++ % vrm_ctx::kBatchEvents];
    HIP_TRY(hipEventRecord(ev, ctx->stream));
    for (uint32_t i = 0; i < n; ++i) {
        vrm_image *img = targets[i];
        img->drawn_batch = ev;
        img->has_draw = true;
        img->copy_valid = false;
        img->copy_pending = false;
    }
    return VRM_OK;
}

extern "C" int vrm_draw_batch(vrm_ctx *ctx, const vrm_camera_uniform *cameras, uint32_t n, vrm_image *const *targets)
{
    return vrm_guard("vrm_draw_batch", VRM_ERR_INVALID, [&]() -> int { return draw_batch_impl(ctx, cameras, n, targets, false); });
}

extern "C" int vrm_draw_batch_static(vrm_ctx *ctx, const vrm_camera_uniform *cameras, uint32_t n, vrm_image *const *targets)
{
    return draw_batch_impl(ctx, cameras, n, targets, true); // nothing in this path allocates or throws
}

extern "C" int vrm_draw(vrm_ctx *ctx, const vrm_camera_uniform *camera, vrm_image *target)
{
    return vrm_draw_batch(ctx, camera, 1, &target);
}

// ---- sort-last ----------------------------------------------------------------------

extern "C" int vrm_draw_partial(vrm_ctx *ctx, const vrm_camera_uniform *cameras, uint32_t n, uint32_t width, uint32_t height,
                                float *rgba_f32_device)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if ((rc = grid_ready(ctx, false))) return rc;
    REQUIRE(cameras != nullptr && rgba_f32_device != nullptr, "null argument");
    REQUIRE(n > 0 && n <= kViewRing && n <= 65535, "batch size %u out of range (1..%u)", n, kViewRing);
    REQUIRE(width > 0 && height > 0 && width <= 65536 && height <= 65536, "bad image size %u x %u", width, height);
    if (ctx->views_cap + n > kViewRing) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->views_cap = 0;
    }
    const uint32_t slot = ctx->views_cap;
    ctx->views_cap += n;
    for (uint32_t i = 0; i < n; ++i) {
        VrmViewDesc &d = ctx->views_host[slot + i];
        memset(&d, 0, sizeof d);
        memcpy(d.ivp, cameras[i].inverse_view_proj, 64);
        memcpy(d.pos, cameras[i].position, 16);
        d.partial_rgba = rgba_f32_device + (size_t)i * width * height * 4;
    }
    HIP_TRY(hipMemcpyAsync(ctx->views_dev + slot, ctx->views_host + slot, sizeof(VrmViewDesc) * n, hipMemcpyHostToDevice, ctx->stream));
    VrmMarchParams p;
    memset(&p, 0, sizeof p);
    fill_params(ctx, &p, width, height, (size_t)width * 4);
    p.any_far = any_far_view(p, cameras, n);
    p.rays = rays_for_batch(ctx, p, cameras, n);
    p.views = ctx->views_dev + slot;
    HIP_TRY(vrm_launch_march(p, n, (int)ctx->opt_kernel, false, ctx->opt_skip != 0, ctx->stream));
    return VRM_OK;
}

extern "C" int vrm_composite(vrm_ctx *ctx, const vrm_camera_uniform *camera, const float *partials_device, uint32_t n_slabs,
                             uint32_t width, uint32_t height, uint32_t row0, uint32_t rows, uint8_t *out_rgba8_device,
                             float *out_f32_device)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(camera != nullptr && partials_device != nullptr, "null argument");
    REQUIRE(out_rgba8_device != nullptr || out_f32_device != nullptr, "no output buffer");
    REQUIRE(n_slabs > 0 && n_slabs <= 4096, "slab count %u out of range", n_slabs);
    REQUIRE(width > 0 && height > 0 && rows > 0 && row0 < height && rows <= height - row0, "rows [%u, %u) outside the %u x %u image",
            row0, row0 + rows, width, height);
    VrmCompositeView cv;
    memcpy(cv.ivp, camera->inverse_view_proj, 64);
    HIP_TRY(vrm_launch_composite(partials_device, n_slabs, cv, width, height, row0, rows, (uint32_t *)out_rgba8_device, out_f32_device,
                                 ctx->stream));
    return VRM_OK;
}

extern "C" int vrm_count_reference_samples(vrm_ctx *ctx, const vrm_camera_uniform *cameras, uint32_t n, uint32_t width,
                                           uint32_t height, uint64_t *out_total)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if ((rc = grid_ready(ctx, false))) return rc;
    REQUIRE(cameras != nullptr && out_total != nullptr, "null argument");
    REQUIRE(n > 0 && n <= kViewRing && n <= 65535, "batch size %u out of range (1..%u)", n, kViewRing);
    REQUIRE(width > 0 && height > 0 && width <= 65536 && height <= 65536, "bad image size %u x %u", width, height);
    if (ctx->views_cap + n > kViewRing) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->views_cap = 0;
    }
    const uint32_t slot = ctx->views_cap;
    ctx->views_cap += n;
    for (uint32_t i = 0; i < n; ++i) {
        VrmViewDesc &d = ctx->views_host[slot + i];
        memset(&d, 0, sizeof d);
        memcpy(d.ivp, cameras[i].inverse_view_proj, 64);
        memcpy(d.pos, cameras[i].position, 16);
    }
    HIP_TRY(hipMemcpyAsync(ctx->views_dev + slot, ctx->views_host + slot, sizeof(VrmViewDesc) * n, hipMemcpyHostToDevice, ctx->stream));
    VrmMarchParams p;
    memset(&p, 0, sizeof p);
    fill_params(ctx, &p, width, height, (size_t)width * 4);
    p.views = ctx->views_dev + slot;
    unsigned long long *cnt = reinterpret_cast<unsigned long long *>(ctx->flag_dev + 2); // 8-byte aligned word of the 256 B scratch
    HIP_TRY(hipMemsetAsync(cnt, 0, 8, ctx->stream));
    HIP_TRY(vrm_launch_count_samples(p, n, cnt, ctx->stream));
    unsigned long long host = 0;
    HIP_TRY(hipMemcpyAsync(&host, cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out_total = host;
    return VRM_OK;
}

extern "C" int vrm_diag_read_counters(vrm_ctx *ctx, uint64_t out[8], int reset)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(out != nullptr, "null argument");
    HIP_TRY(hipMemcpyAsync(out, ctx->flag_dev + 16, 64, hipMemcpyDeviceToHost, ctx->stream));
    if (reset) HIP_TRY(hipMemsetAsync(ctx->flag_dev + 16, 0, 64, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return VRM_OK;
}

extern "C" int vrm_synchronize(vrm_ctx *ctx)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
    return VRM_OK;
}

// ---- options / profiling ------------------------------------------------------------------------

// ---- the one collective of the view-partitioned export, inside the library ----------------------------
// RCCL is bound at first use (dlopen), so the library has no link-time dependency on it and a process that already holds
// RCCL (PyTorch) shares that copy.  Only the four entry points used here are resolved.
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;                                      // ncclGetUniqueId(ncclUniqueId*)
    int (*CommInitRank)(void **, int, vrm_comm_id, int) = nullptr;             // ncclCommInitRank(comm*, n, id BY VALUE, rank)
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};
Rccl &rccl()
{
    static Rccl r = [] {
        Rccl q;
        // VRM_RCCL_LIB names the library to bind instead of the usual candidates (a site's own build; the CPU test of the
        // "no RCCL on this machine" path points it at nothing)
        const char *forced = getenv("VRM_RCCL_LIB");
        std::string last;
        if (forced && *forced) {
            q.lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
            if (!q.lib) { const char *e = dlerror(); last = e ? e : "dlopen failed"; } // (dlerror() clears the message it returns: one call)
        } else {
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                q.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (q.lib) break;
                const char *e = dlerror();
                last = e ? e : "dlopen failed";
            }
        }
        if (!q.lib) {
            q.why = "RCCL not found (" + last + ")";
            return q;
        }
        q.GetUniqueId = reinterpret_cast<decltype(q.GetUniqueId)>(dlsym(q.lib, "ncclGetUniqueId"));
        q.CommInitRank = reinterpret_cast<decltype(q.CommInitRank)>(dlsym(q.lib, "ncclCommInitRank"));
        q.Broadcast = reinterpret_cast<decltype(q.Broadcast)>(dlsym(q.lib, "ncclBroadcast"));
        q.CommDestroy = reinterpret_cast<decltype(q.CommDestroy)>(dlsym(q.lib, "ncclCommDestroy"));
        q.GetErrorString = reinterpret_cast<decltype(q.GetErrorString)>(dlsym(q.lib, "ncclGetErrorString"));
        if (!q.GetUniqueId || !q.CommInitRank || !q.Broadcast || !q.CommDestroy) {
            q.why = "RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclBroadcast / ncclCommDestroy";
            q.lib = nullptr;
        }
        return q;
    }();
    return r;
}
int rccl_fail(const char *what, int code)
{
    Rccl &r = rccl();
    vrm_set_error("%s failed: %s (RCCL result %d)", what, r.GetErrorString ? r.GetErrorString(code) : "?", code);
    return VRM_ERR_HIP;
}
} // namespace

extern "C" int vrm_comm_unique_id(vrm_comm_id *id)
{
    REQUIRE(id != nullptr, "null id");
    return vrm_guard("vrm_comm_unique_id", VRM_ERR_NO_DEVICE, [&]() -> int { // (binding RCCL builds strings: nothing throws across the ABI)
        Rccl &r = rccl();
        if (!r.lib) {
            vrm_set_error("%s", r.why.c_str());
            return VRM_ERR_NO_DEVICE;
        }
        static_assert(sizeof(vrm_comm_id) == 128, "ncclUniqueId is 128 bytes");
        const int e = r.GetUniqueId(id);
        return e == 0 ? VRM_OK : rccl_fail("ncclGetUniqueId", e);
    });
}

extern "C" int vrm_comm_init(vrm_ctx *ctx, const vrm_comm_id *id, int rank, int n_ranks)
{
    int rc = bind(ctx);
    if (rc) return rc;
    REQUIRE(id != nullptr, "null id");
    REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad rank %d of %d", rank, n_ranks);
    REQUIRE(ctx->comm == nullptr, "this context already has a communicator (vrm_comm_destroy first)");
    return vrm_guard("vrm_comm_init", VRM_ERR_NO_DEVICE, [&]() -> int {
        Rccl &r = rccl();
        if (!r.lib) {
            vrm_set_error("%s", r.why.c_str());
            return VRM_ERR_NO_DEVICE;
        }
        void *comm = nullptr;
        const int e = r.CommInitRank(&comm, n_ranks, *id, rank); // collective: every rank of the job calls it
        if (e != 0) return rccl_fail("ncclCommInitRank", e);
        ctx->comm = comm;
        ctx->comm_rank = rank;
        ctx->comm_size = n_ranks;
        return VRM_OK;
    });
}

extern "C" int vrm_grid_broadcast(vrm_ctx *ctx, int root)
{
    int rc = bind(ctx);
    if (rc) return rc;
    if (ctx->comm == nullptr) {
        vrm_set_error("no communicator: call vrm_comm_init on every rank first");
        return VRM_ERR_STATE;
    }
    REQUIRE(root >= 0 && root < ctx->comm_size, "root %d outside the %d ranks", root, ctx->comm_size);
    REQUIRE(ctx->abricks != nullptr && ctx->vox_words != 0, "no grid");
    // what travels is the RGBA8 form every rank ingests: the root needs its voxels there, a receiver only the storage (a grid it
    // resized for this and never filled is not filled with default voxels first)
    if (ctx->comm_rank != root && ctx->lazy_default) rc = grid_rgba_storage(ctx);
    else rc = grid_ready(ctx, true);
    if (rc) return rc;
    REQUIRE(ctx->gdim[2] == ctx->dim[2], "a slab-partitioned grid is not replicated: nothing to broadcast");
    // every rank must hold a grid of the same dimensions (vrm_grid_resize); the root's voxels replace the others' in place
    const int e = rccl().Broadcast(ctx->vox, ctx->vox, ctx->vox_words * 4, /* ncclUint8 */ 1, root, ctx->comm, ctx->stream);
    if (e != 0) return rccl_fail("ncclBroadcast", e);
    if (ctx->comm_rank == root) return vrm_check_hip(hipStreamSynchronize(ctx->stream), "hipStreamSynchronize");
    return grid_finish_upload(ctx); // bricks, packed alpha, occupancy pyramid, scalar detection of the received voxels
}

extern "C" int vrm_comm_destroy(vrm_ctx *ctx)
{
    REQUIRE(ctx != nullptr, "null context");
    if (ctx->comm == nullptr) return VRM_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    const int e = rccl().CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_size = 0;
    return e == 0 ? VRM_OK : rccl_fail("ncclCommDestroy", e);
}

extern "C" int vrm_set_option(vrm_ctx *ctx, int option, int64_t value)
{
    REQUIRE(ctx != nullptr, "null context");
    switch (option) {
    case VRM_OPT_COMPACT_SCALAR: ctx->opt_compact = value; break;
    case VRM_OPT_KERNEL: ctx->opt_kernel = value; break;
    case VRM_OPT_SKIP_EMPTY: ctx->opt_skip = value; break;
    case VRM_OPT_PROFILE: ctx->opt_profile = value; break;
    case VRM_OPT_DIAGNOSTICS: ctx->opt_diag = value; break;
    case VRM_OPT_CHUNK_STEPS: ctx->opt_chunk = value < 0 ? 0 : (value > 256 ? 256 : value); break;
    case VRM_OPT_PACKED_ALPHA: ctx->opt_packed = value != 0 ? 1 : 0; break;
    case VRM_OPT_RAYS_PER_LANE: ctx->opt_rays = value == 1 || value == 2 ? value : 0; break;
    default: vrm_set_error("unknown option %d", option); return VRM_ERR_INVALID;
    }
    return VRM_OK;
}

extern "C" int vrm_get_option(vrm_ctx *ctx, int option, int64_t *value)
{
    REQUIRE(ctx != nullptr && value != nullptr, "null argument");
    switch (option) {
    case VRM_OPT_KERNEL: *value = ctx->opt_kernel; break;
    case VRM_OPT_SKIP_EMPTY: *value = ctx->opt_skip; break;
    case VRM_OPT_PROFILE: *value = ctx->opt_profile; break;
    case VRM_OPT_DIAGNOSTICS: *value = ctx->opt_diag; break;
    case VRM_OPT_CHUNK_STEPS: *value = ctx->opt_chunk; break;
    case VRM_OPT_PACKED_ALPHA: *value = ctx->opt_packed; break;
    case VRM_OPT_COMPACT_SCALAR: *value = ctx->opt_compact; break;
    case VRM_OPT_RAYS_PER_LANE: *value = ctx->opt_rays; break;
    default: vrm_set_error("unknown option %d", option); return VRM_ERR_INVALID;
    }
    return VRM_OK;
}

extern "C" int vrm_profile_read(vrm_ctx *ctx, double *kernel_ms, uint64_t *launches, int reset)
{
    int rc = bind(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (size_t i = 0; i < ctx->prof_used; ++i) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_events[i].first, ctx->prof_events[i].second));
        ctx->prof_ms += ms;
        ctx->prof_launches++;
    }
    ctx->prof_used = 0;
    if (kernel_ms) *kernel_ms = ctx->prof_ms;
    if (launches) *launches = ctx->prof_launches;
    if (reset) {
        ctx->prof_ms = 0.0;
        ctx->prof_launches = 0;
    }
    return VRM_OK;
}
