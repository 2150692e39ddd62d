/*
 * vrm.h -- C ABI of the MI355X ray-marching path ("vrm" = volume ray marcher).
 *
 * This is the drop-in boundary for the one hot path of bennetde/volume-renderer: everything the
 * Rust host (`State`, src/state.rs) asks of `RayMarcher`, `VoxelGrid`, `Texture3D` and
 * `Screenshotter` is one call here.  The reference has no FFI of its own; each entry point cites
 * the Rust interface it replaces (paths relative to the reference's src/).  INTEGRATION.md shows the
 * `extern "C"` block a `vrm-sys` crate would declare.
 *
 * Conventions
 *   - plain pointers and sizes only; opaque handles; no C++/torch types.
 *   - every function returns 0 on success or a negative vrm_status; vrm_last_error() returns a
 *     thread-local message for the last failure.  Nothing aborts across the boundary (the reference
 *     panics / unwrap()s: grid.rs:209, screenshot.rs:81, state.rs:273).
 *   - a vrm_ctx is bound to one GPU and is not thread-safe (the reference is single-threaded:
 *     Rc<BindGroup>, ray_marcher/mod.rs:21).  Different contexts may be used from different threads.
 *   - host pointers passed in are copied before the call returns unless stated otherwise.
 *   - the library fails loudly (VRM_ERR_NO_DEVICE) when no HIP device is present; there is no CPU
 *     fallback of any kind.
 */
#ifndef VRM_H
#define VRM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VRM_API __attribute__((visibility("default")))

typedef enum vrm_status {
    VRM_OK = 0,
    VRM_ERR_INVALID = -1,   /* bad argument (null pointer, zero size, size mismatch) */
    VRM_ERR_NO_DEVICE = -2, /* no usable HIP device */
    VRM_ERR_HIP = -3,       /* a HIP runtime call failed; see vrm_last_error() */
    VRM_ERR_IO = -4,        /* file could not be read / written / parsed */
    VRM_ERR_STATE = -5      /* call made in the wrong state (e.g. map before readback) */
} vrm_status;

/* ---- byte-exact uniform structs ---------------------------------------------------------- */

/* CameraUniform, camera.rs:42-50 / raymarcher.wgsl:2-6.  144 bytes, matrices column-major. */
typedef struct vrm_camera_uniform {
    float position[4];          /* xyz, w = 0 */
    float view_proj[16];        /* unused by the march, carried for layout parity */
    float inverse_view_proj[16];
} vrm_camera_uniform;

/* RaymarchTransferFunctionColors, voxel/grid.rs:34-41 / raymarcher.wgsl:22-27.  64 bytes. */
typedef struct vrm_transfer_function {
    float color_a[4];
    float color_b[4];
    float color_c[4];
    uint32_t use_transfer_function[4]; /* [0] != 0 => on */
} vrm_transfer_function;

/* VoxelGridUniform, voxel/grid.rs:24-32 / raymarcher.wgsl:14-20.  64 bytes. */
typedef struct vrm_grid_uniform {
    uint32_t dimensions[4]; /* X, Y, Z, X */
    float box_min[4];
    float box_size[4];
    float buffer[4];        /* attenuation x4 */
} vrm_grid_uniform;

/* SurfaceConfiguration fields RayMarcher::new / Screenshotter::new read (state.rs:72-84). */
typedef struct vrm_config {
    int32_t device_id; /* HIP device ordinal */
    uint32_t width;    /* default target size, used by vrm_export_sphere */
    uint32_t height;
} vrm_config;

typedef struct vrm_ctx vrm_ctx;     /* RayMarcher + its VoxelGrid (ray_marcher/mod.rs:18-23) */
typedef struct vrm_image vrm_image; /* render target + Screenshotter buffer (screenshot.rs:6-8) */

/* ---- RayMarcher --------------------------------------------------------------------------- */

/* RayMarcher::new (ray_marcher/mod.rs:26-111).  Creates the context with the reference's start-up
 * grid: 3x3x3, every voxel [0,0,0,255], attenuation 1.0, transfer function off (mod.rs:29,
 * voxel.rs:13-16, grid.rs:197-198). */
VRM_API int vrm_create(const vrm_config *config, vrm_ctx **out_ctx);
VRM_API void vrm_destroy(vrm_ctx *ctx);

/* RayMarcher::draw (ray_marcher/mod.rs:113-141).  Enqueues one view into `target` (cleared to
 * transparent, REPLACE blend => every pixel is overwritten) and returns without waiting, like the
 * unsubmitted CommandBuffer the reference returns; completion is observed by vrm_image_map or
 * vrm_synchronize. */
VRM_API int vrm_draw(vrm_ctx *ctx, const vrm_camera_uniform *camera, vrm_image *target);

/* n independent draws in one launch (the export loop state.rs:227-239 + 382-397 issues them one
 * frame at a time).  cameras[i] renders into targets[i]; all targets must have the same size. */
VRM_API int vrm_draw_batch(vrm_ctx *ctx, const vrm_camera_uniform *cameras, uint32_t n, vrm_image *const *targets);

/* vrm_draw_batch for callers that cannot allocate: tight serving loops, harnesses that forbid hipMalloc.  (NOT a graph-capture
 * entry point: the ABI exposes no stream to capture on, the descriptors are staged through a pinned ring whose slots later
 * calls reuse, and once every 4096 views the call waits for the stream before the ring wraps.)  NO COUNTERPART IN THE REFERENCE (wgpu allocates its command buffers and staging per frame).  Same arguments,
 * same kernels, same image as vrm_draw_batch; the difference is where the march kernel's scratch -- the per-view
 * descriptors (inverse view-projection, position, output pointers; 120 B per view) -- lives: in a file-scope device array
 * of the library (30 KiB per GPU, held until the process ends) instead of the ring vrm_create allocated.  The call itself
 * performs no hipMalloc / hipFree / event creation and no host allocation.  Limits: n <= 256 views per call (larger batches
 * are VRM_ERR_INVALID; split them); VRM_OPT_DIAGNOSTICS and VRM_OPT_PROFILE are ignored (both allocate); VRM_ERR_STATE for a
 * volume whose RGBA copies this draw needs and are not there (a compact scalar volume with RGBA-staging kernel options, or a
 * grid that was resized and not yet uploaded to or drawn: vrm_draw_batch materialises those).  Every call on a
 * GPU shares that GPU's array: calls are stream-ordered within one context, but calls from DIFFERENT contexts on the same
 * GPU that overlap in time race on it -- serialise them (vrm_synchronize) or use vrm_draw_batch. */
VRM_API int vrm_draw_batch_static(vrm_ctx *ctx, const vrm_camera_uniform *cameras, uint32_t n, vrm_image *const *targets);

/* device.poll(Maintain::wait()) (screenshot.rs:62): block until everything enqueued on ctx is done. */
VRM_API int vrm_synchronize(vrm_ctx *ctx);

/* ---- VoxelGrid / Texture3D ---------------------------------------------------------------- */

/* VoxelGrid::new (voxel/grid.rs:44-200) as the loaders call it (dat.rs:57, netcdf.rs:47): new
 * X*Y*Z grid of default voxels; attenuation resets to 1.0 (grid.rs:100,197); the transfer-function
 * struct resets to its defaults (grid.rs:198).  The default voxels [0,0,0,255] are a promise: every loader overwrites them
 * at once (dat.rs:21, netcdf.rs:47), so the call allocates about 1 B/voxel and the 8 B/voxel of RGBA storage with the fill
 * come when something looks at the grid before an upload has replaced it (a draw, a download) -- a scalar upload
 * (vrm_grid_upload_scalar) then never allocates them.  The 3x3x3 grid vrm_create makes is filled at once. */
VRM_API int vrm_grid_resize(vrm_ctx *ctx, uint32_t x, uint32_t y, uint32_t z);

/* set_color over the whole grid + update_buffer (grid.rs:202-238, Texture3D::from_image
 * texture_3d.rs:22-80): RGBA8 voxels, x fastest (idx = x + X*(y + Y*z)), nbytes must be 4*X*Y*Z. */
VRM_API int vrm_grid_upload_rgba8(vrm_ctx *ctx, const uint8_t *host_voxels, size_t nbytes);
/* same, source already in this GPU's memory (e.g. after an RCCL broadcast); enqueued on ctx's stream. */
VRM_API int vrm_grid_upload_rgba8_device(vrm_ctx *ctx, const void *device_voxels, size_t nbytes);

/* loaders/netcdf.rs:45-61: f32 `color(c, z, y, x)`; each value becomes `(v*255.0) as u8`
 * (truncate, saturate, NaN -> 0).  channels = 4, or 1 for a scalar field replicated to r=g=b=a. */
VRM_API int vrm_grid_upload_f32(vrm_ctx *ctx, const float *host_czyx, uint32_t channels);
VRM_API int vrm_grid_upload_f32_device(vrm_ctx *ctx, const void *device_czyx, uint32_t channels);

/* loaders/dat.rs:58-76 quantisation: raw UCHAR (bytes_per_sample 1) or USHORT combined big-endian
 * (bytes_per_sample 2); voxel = [b,b,b,b].  With the default options (VRM_OPT_COMPACT_SCALAR, the packed march) the packed
 * alpha bricks and the occupancy pyramid are built straight from the raw samples: the ingest peaks at the raw samples
 * + 1 B/voxel on the device, never the 9 B/voxel of the RGBA form. */
VRM_API int vrm_grid_upload_scalar(vrm_ctx *ctx, const uint8_t *host_raw, size_t nbytes, uint32_t bytes_per_sample);

/* Index<UVec3> over the whole grid (grid.rs:249-263): copies the RGBA8 voxels back, x fastest. */
VRM_API int vrm_grid_download_rgba8(vrm_ctx *ctx, uint8_t *host_voxels, size_t nbytes);

/* voxel_grid.attenuation + update_voxel_grid_buffer (grid.rs:20,240-242; state.rs:365-368) */
VRM_API int vrm_set_attenuation(vrm_ctx *ctx, float attenuation);
/* transfer_function_colors + update_transfer_function_buffer (grid.rs:21,244-246; state.rs:370-374) */
VRM_API int vrm_set_transfer_function(vrm_ctx *ctx, const vrm_transfer_function *tf);
/* the VoxelGridUniform the shader would see (grid.rs:265-280) */
VRM_API int vrm_get_grid_uniform(vrm_ctx *ctx, vrm_grid_uniform *out);

/* ---- Screenshotter ------------------------------------------------------------------------ */

/* Screenshotter::new (screenshot.rs:12-27) plus the surface texture it copies from: an RGBA8
 * (non-sRGB, state.rs:78) target of w*h pixels, device + pinned host copy, row pitch
 * ceil(4w/256)*256 bytes (screenshot.rs:94-96). */
VRM_API int vrm_image_create(vrm_ctx *ctx, uint32_t width, uint32_t height, vrm_image **out_image);
VRM_API void vrm_image_destroy(vrm_image *image);
/* Screenshotter::screenshot (screenshot.rs:30-54): enqueue the device->host copy behind the draws. */
VRM_API int vrm_image_readback_async(vrm_image *image);
/* map_async + poll(wait) + get_mapped_range (screenshot.rs:58-65): blocks until the copy landed.
 * *rgba8 stays valid until vrm_image_unmap; rows are *pitch bytes apart. */
VRM_API int vrm_image_map(vrm_image *image, const uint8_t **rgba8, size_t *pitch);
VRM_API int vrm_image_unmap(vrm_image *image); /* screenshot.rs:91 */
/* save_screenshot_to_disk (screenshot.rs:57-92): readback if needed, strip row padding, write an
 * RGBA8 PNG (premultiplied colour + alpha, as rendered). */
VRM_API int vrm_image_save_png(vrm_image *image, const char *path);

/* ---- camera + export loop (host side of the path) ------------------------------------------ */

/* CameraUniform::update_view_proj (camera.rs:61-67) for a camera at `position` aimed by
 * Transform::look_to(target, +Y) (transform.rs:59-63), fovy 45 deg, near 0.1, far 1000 (camera.rs:19-21).
 * basis9 (may be NULL) receives right, up, front as written to cameras.json. */
VRM_API int vrm_camera_look_at(const float position[3], const float target[3], float aspect, vrm_camera_uniform *out,
                               float basis9[9]);
/* CameraSphereController::get_position_on_sphere (camera_sphere_controller.rs:79-89) */
VRM_API int vrm_sphere_position(uint32_t ix, uint32_t iy, uint32_t x_div, uint32_t y_div, float radius,
                                const float origin[3], float out_position[3]);

/* "Screenshot All" (sphere_screenshot_manager.rs:23-73, state.rs:318-322,391-397): renders the
 * x_div*(y_div-1) views (y outer from 1, x inner), writes `<out_dir>/<ix>-<iy>.png` and
 * `<out_dir>/cameras.json`.  Of that list this context renders views i with (i + i / x_div) % n_parts == part -- interleaved, the interleave rotated by one from one
 * row of the sphere to the next, so that a part does not get the same azimuths in every row (plain i % n_parts is 10 % out of
 * balance at 8 parts of the 16-column export: the axis-aligned views are the cheap ones)
 * (view partitioning over GPUs/ranks); only part 0 writes cameras.json.  Uses ctx's default size. */
VRM_API int vrm_export_sphere(vrm_ctx *ctx, uint32_t x_div, uint32_t y_div, float radius, const float origin[3],
                              const char *out_dir, uint32_t part, uint32_t n_parts);

/* loaders/dat.rs:10-82: parse `ObjectFileName/Resolution/Format`, read the RAW next to it, upload. */
VRM_API int vrm_load_dat(vrm_ctx *ctx, const char *dat_path);

/* ---- NetCDF interchange ------------------------------------------------------ */

/* loaders::netcdf::open_voxel_grid (loaders/netcdf.rs:34-69): read variable `color(c,z,y,x)` (f32, c = 4), make a
 * fresh x*y*z grid (attenuation and TF reset, netcdf.rs:47) and quantise `(v*255.0) as u8` per channel (netcdf.rs:54-57,
 * on the GPU).  Containers: NetCDF-4 = HDF5 -- what the reference's netcdf crate (loaders/netcdf.rs:10,35) and
 * scripts/diffdvr.py:278 write; the subset csrc/vrm_hdf5.cpp lists: contiguous, chunked and single-chunk f32 datasets,
 * deflate / shuffle / fletcher32 -- and netCDF classic (CDF-1, CDF-2, CDF-5).  HDF5 features outside the subset (dense
 * link storage, newer chunk indices) and sizes that do not fit the file are VRM_ERR_IO with a message. */
VRM_API int vrm_load_netcdf(vrm_ctx *ctx, const char *path);
/* loaders::netcdf::write_voxel_grid (loaders/netcdf.rs:8-31): dimensions c=4,z,y,x and `color` = voxel/255.0 as f32,
 * written as a classic file (CDF-1 below 2 GiB, CDF-2 below 4 GiB, CDF-5 beyond). */
VRM_API int vrm_write_netcdf(vrm_ctx *ctx, const char *path);
/* The same volume in the container the reference's own writer makes (`netcdf::create`, netcdf.rs:10): NetCDF-4 = HDF5,
 * in the oldest dialect of that format (superblock 0, symbol-table root group, contiguous little-endian `color`, the
 * dimensions as dimension-scale datasets with netcdf-c's CLASS / NAME / _Netcdf4Dimid attributes, `color` with
 * _Netcdf4Coordinates).  Checked against the HDF5 library (libhdf5 1.10.6's h5dump reads it: tests/test_hdf5_golden_cpu.py), this library's reader and a second restatement of the
 * specification (tests/hdf5_builder.py), not against netcdf-c, which does not exist in the build environment. */
VRM_API int vrm_write_netcdf4(vrm_ctx *ctx, const char *path);
/* The container alone, host only (no context, no GPU): with color_czyx == NULL a size query that sets x, y, z;
 * otherwise 4*x*y*z floats in file order (c slowest, x fastest) converted to host byte order.  `color` stored as another
 * numeric type (double, integers) is handed over as (float)value, the conversion of netcdf-c's nc_get_vara_float behind the
 * reference's get::<f32> (netcdf.rs:45; a double beyond the range of f32 is NC_ERANGE there and VRM_ERR_IO here, text is refused). */
VRM_API int vrm_netcdf_read_color(const char *path, uint32_t *x, uint32_t *y, uint32_t *z, float *color_czyx,
                                  size_t capacity_floats);
VRM_API int vrm_netcdf_write_color(const char *path, uint32_t x, uint32_t y, uint32_t z, const float *color_czyx);
VRM_API int vrm_netcdf_write_color_nc4(const char *path, uint32_t x, uint32_t y, uint32_t z, const float *color_czyx);

/* ---- volume broadcast between the ranks of a view-partitioned export ---------------------
 * No reference counterpart (the reference is one process on one adapter).  north_star: "the volume is broadcast once
 * with RCCL over xGMI" -- for host programs that are not PyTorch (the Rust caller): one context per GPU and process,
 * rank 0 makes an id and hands its 128 bytes to the other ranks by its own means (file, socket, MPI), every rank calls
 * vrm_comm_init (a collective), every rank makes a grid of the same dimensions (vrm_grid_resize), the root uploads,
 * everybody calls vrm_grid_broadcast: the root's RGBA8 voxels arrive in every context's grid (ncclBroadcast on the
 * context's stream) and the receivers rebuild their bricks and occupancy pyramid.  RCCL is bound at first use (dlopen of
 * librccl), so nothing else in the library depends on it; VRM_ERR_NO_DEVICE if it cannot be found.  The attenuation
 * and the transfer function are 4 and 64 bytes: set them on every rank. */
typedef struct vrm_comm_id {
    uint8_t bytes[128]; /* an ncclUniqueId */
} vrm_comm_id;
VRM_API int vrm_comm_unique_id(vrm_comm_id *id);
VRM_API int vrm_comm_init(vrm_ctx *ctx, const vrm_comm_id *id, int rank, int n_ranks);
VRM_API int vrm_grid_broadcast(vrm_ctx *ctx, int root);
/* also done by vrm_destroy */
VRM_API int vrm_comm_destroy(vrm_ctx *ctx);

/* ---- sort-last rendering of a z-slab-partitioned volume (no reference counterpart: ---------
 * ---- the reference holds the whole volume in one 3D texture, texture_3d.rs:22-80, state.rs:59) ---------- */

/* VoxelGrid::new for ONE SLAB of an x*y*z volume: this context will own the samples whose position lies in the
 * slices [z_begin, z_end) and stores slices [max(z_begin-1,0), min(z_end+1,z)) -- one ghost slice either side, what
 * the trilinear filter of an owned sample can touch.  Box, step lattice, t_min/t_max stay those of the WHOLE volume
 * (vrm_get_grid_uniform reports x,y,z), so every rank marches the same global sample lattice and each sample has
 * exactly one owner (half-open world-space bounds, the same f32 expression on every rank).  The upload calls then
 * take the STORED slices only (x*y*(stored) voxels, slice order ascending).  Resets attenuation/TF like
 * vrm_grid_resize. */
VRM_API int vrm_grid_resize_slab(vrm_ctx *ctx, uint32_t x, uint32_t y, uint32_t z, uint32_t z_begin, uint32_t z_end);
/* device bytes the current grid holds (voxel copies, bricks, occupancy pyramid): 9 B/voxel for an RGBA volume, about 1 B/voxel
 * for a scalar one in its compact form (VRM_OPT_COMPACT_SCALAR).  No reference counterpart (a wgpu texture's size is implied). */
VRM_API int vrm_grid_resident_bytes(vrm_ctx *ctx, size_t *bytes);
/* stored slice range and owned slice range of the current grid ([0,z) twice for an unpartitioned grid) */
VRM_API int vrm_grid_get_slab(vrm_ctx *ctx, uint32_t *z_store_begin, uint32_t *z_store_end, uint32_t *z_begin, uint32_t *z_end);

/* RayMarcher::draw of n views into caller-owned DEVICE memory as premultiplied f32 RGBA (n * height * width * 4
 * floats, rows top to bottom, no padding): the accumulators of wgsl:193 before the Rgba8Unorm store.  For a slab
 * context this is the slab's partial image.  Enqueue only, on the context's stream: vrm_synchronize before another
 * stream or library (RCCL) reads it. */
VRM_API int vrm_draw_partial(vrm_ctx *ctx, const vrm_camera_uniform *cameras, uint32_t n, uint32_t width, uint32_t height,
                             float *rgba_f32_device);

/* "Over"-composite n_slabs partial images of the rows [row0, row0+rows) of one view -- partials_device[slab][row]
 * [x][4] floats, slab 0 lowest in z -- in the order the pixel's ray visits the slabs (ascending iff rd.z >= 0) and
 * store the rows like the single-GPU path does: RGBA8 (rows*width*4 bytes, round(clamp(v,0,1)*255)) and/or f32
 * (either output may be NULL).  Enqueue only. */
VRM_API int vrm_composite(vrm_ctx *ctx, const vrm_camera_uniform *camera, const float *partials_device, uint32_t n_slabs,
                          uint32_t width, uint32_t height, uint32_t row0, uint32_t rows, uint8_t *out_rgba8_device,
                          float *out_f32_device);

/* ---- evaluation (the DiffDVR loop around the path) ---------------------------- */

/* compare_to_netcdf_rmse (compare/netcdf.rs:8-75): RMSE between the loaded grid and a `color(c,z,y,x)` f32
 * volume of the same dimensions (4 channels, channel-major, x fastest -- the NetCDF variable as stored).  Per
 * voxel (alpha/255 - color[3])^2 and, unless density_only, the transfer-function colour of alpha (the context's
 * three colours, breakpoint 0.5, whether or not the TF is switched on, netcdf.rs:41-61) against color[0..2];
 * mean over voxels * (1 or 4), square root.  Terms are formed in f32 as the reference forms them; the sum is f64
 * (the reference's one sequential f32 sum loses digits beyond ~10^7 voxels, see tests). */
VRM_API int vrm_compare_volume_f32(vrm_ctx *ctx, const float *color_czyx, int density_only, float *out_rmse);
VRM_API int vrm_compare_volume_f32_device(vrm_ctx *ctx, const float *color_czyx_device, int density_only, float *out_rmse);

/* skimage.metrics.mean_squared_error on two uint8 images (scripts/similarity.py:59-61) is SSD / count; these
 * return the exact integer sum of squared differences, of two rendered images (device side, same size; count =
 * 4*w*h) or of two host arrays of n bytes.  rmse = sqrt(mse), psnr = 10 log10(255^2 / mse) (similarity.py:62-63). */
VRM_API int vrm_image_ssd(vrm_image *a, vrm_image *b, uint64_t *out_ssd, uint64_t *out_count);
VRM_API int vrm_ssd_u8(vrm_ctx *ctx, const uint8_t *a, const uint8_t *b, size_t n, uint64_t *out_ssd);

/* skimage.metrics.structural_similarity(a, b, data_range=data_range, channel_axis=-1) on two uint8 images of
 * width*height*channels interleaved host bytes (scripts/similarity.py:64-65; the caller forms data_range as the
 * script does).  The scikit-image defaults are restated: 7x7 uniform window, K1 = 0.01, K2 = 0.03, sample
 * covariance, f64, 3-pixel border cropped, mean per channel and then over channels.  scikit-image is a dependency
 * of the script that is absent from the reference tree, so this restates its published algorithm (Wang et al. 2004);
 * pinned against scikit-image 0.18.3 itself: 1e-12 on the golden pairs of tests/golden/similarity.  Images smaller than the
 * window are VRM_ERR_INVALID (skimage raises ValueError). */
VRM_API int vrm_ssim_u8(vrm_ctx *ctx, const uint8_t *a, const uint8_t *b, uint32_t width, uint32_t height, uint32_t channels,
                        double data_range, double *out_ssim);

/* skimage.io.imread (similarity.py:59-60) for the files this path writes and reads: 8-bit non-interlaced
 * gray / gray+alpha / RGB / RGBA PNG.  Always sets width, height, channels; with pixels == NULL that is all
 * (size query), otherwise decodes width*height*channels bytes, rows top to bottom, into pixels[capacity]. */
VRM_API int vrm_png_read(const char *path, uint32_t *width, uint32_t *height, uint32_t *channels, uint8_t *pixels,
                         size_t capacity);

/* ---- build extensions (no reference counterpart) --------------------------------------------- */

typedef enum vrm_option {
    VRM_OPT_KERNEL = 0,       /* march kernel: 0 = global gathers (baseline), 1 = per-wave LDS staging, 10 KB tile (default; tiles handed
                                 out through per-XCD queues or one workgroup per tile, whichever measured faster for the case),
                                 2 = the same with an 8 KB tile, 5 waves per SIMD, 3 / 4 = variant 1 with queues / workgroups
                                 forced (tuning, DESIGN.md "Kernels"); the image does not depend on 1..4 */
    VRM_OPT_SKIP_EMPTY = 1,   /* exact empty-space skipping, default 1: chunks whose texels all have alpha == 0 are not
                                 sampled; they contribute exactly 0, so the image is bit-identical to 0 (tests) */
    VRM_OPT_PROFILE = 2,      /* default 0; 1: record hipEvents around every march launch (vrm_profile_read) */
    VRM_OPT_DIAGNOSTICS = 3,  /* default 0; 1: also write per-pixel f32 colour, iteration counts and ray geometry */
    VRM_OPT_CHUNK_STEPS = 4,  /* default 0 = automatic; n: march steps per LDS-staged box (tuning, no effect on the image) */
    VRM_OPT_PACKED_ALPHA = 5, /* 1-channel march (scalar volumes, transfer function) staged from the packed alpha copy of
                                 the volume, one byte per texel: 1 = on (default), 0 = stage the RGBA8 texels instead (tuning; same
                                 image to f32 rounding) */
    VRM_OPT_COMPACT_SCALAR = 6, /* default 1: after an upload that turned out scalar (r == g == b == a in every voxel: DAT files,
                                 1-channel f32) keep only what the 1-channel march reads -- packed alpha bricks + occupancy pyramid,
                                 about 1 B/voxel -- and release the 8 B/voxel of RGBA8 copies; they are rebuilt (exactly) when a
                                 download, a broadcast, the voxel compare or an RGBA-staging kernel option needs them.  0: keep
                                 everything resident (9 B/voxel).  Takes effect at the next upload. */
    VRM_OPT_RAYS_PER_LANE = 7  /* wave tile of the packed scalar march: 0 (default) = automatic -- two rays per lane (16x8-pixel
                                 tiles, one staged box per chunk shared by both rays) for volumes of more than 2^28 voxels when a
                                 pixel is narrower than 0.6 texel in every view of the batch, else one (8x8) --, 1 / 2 = forced (tuning; the image does not
                                 depend on it).  Far views, slab-partitioned volumes and the transfer function keep one. */
} vrm_option;
VRM_API int vrm_set_option(vrm_ctx *ctx, int option, int64_t value);
VRM_API int vrm_get_option(vrm_ctx *ctx, int option, int64_t *value);

/* With VRM_OPT_PROFILE on: total device time (hipEvent, on the stream the kernels were launched
 * on) and number of march launches since the last reset. */
VRM_API int vrm_profile_read(vrm_ctx *ctx, double *kernel_ms, uint64_t *launches, int reset);

/* With VRM_OPT_DIAGNOSTICS on at draw time: premultiplied f32 RGBA (w*h*4 floats) and per-ray loop
 * iteration counts (w*h u32) of the last draw into `image`; either pointer may be NULL. */
VRM_API int vrm_image_read_diagnostics(vrm_image *image, float *rgba_f32, uint32_t *iterations);
/* Same switch: per ray, the number of samples that were actually filtered and blended in the last draw (w*h u32):
 * the in-box count minus the samples in chunks the empty-space skipping passed over (N_exec of SURVEY.md 8d; the
 * iteration count above is the reference-equivalent N_ref and does not depend on skipping). */
VRM_API int vrm_image_read_sample_counts(vrm_image *image, uint32_t *executed);
/* Same switch: 8 floats per pixel -- ray direction xyz, t_min, t_max, number of samples that passed the
 * in-box test (wgsl:218-223), hit flag (wgsl:131), 0 -- so tests can check the geometry bit for bit. */
VRM_API int vrm_image_read_ray_diagnostics(vrm_image *image, float *geom8);

/* Same switch: counters summed over every diagnostics draw since the last reset -- [0] samples filtered and blended (N_exec),
 * [1] of those, samples whose filtered alpha was exactly 0 (they changed nothing; what finer skipping could still save),
 * [2] chunk prologues, [3] chunks staged into LDS and sampled, [4] chunks passed over as empty, [5] deep empty-space probes
 * that found something and were redone as plain chunks, [6] chunks marched with global gathers, [7] box computations repeated
 * with fewer steps because the first box did not fit the LDS tile. */
VRM_API int vrm_diag_read_counters(vrm_ctx *ctx, uint64_t out[8], int reset);

/* Reference-equivalent sample count (the unit of the Gsamples/s metric): the number of
 * iterations the loop of raymarch() (raymarcher.wgsl:139-187) executes, summed over all rays of the n
 * views at width x height, for the current grid's box -- pure geometry on the reference's dt lattice,
 * without the `alpha >= 1.0` exit (which cannot fire while a*attenuation*0.001 < ln 2). */
VRM_API int vrm_count_reference_samples(vrm_ctx *ctx, const vrm_camera_uniform *cameras, uint32_t n, uint32_t width,
                                        uint32_t height, uint64_t *out_total);

VRM_API const char *vrm_last_error(void);
VRM_API const char *vrm_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VRM_H */
