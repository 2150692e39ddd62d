// vrm_device.h -- structs shared between the C-ABI host code (vrm_api.cpp) and the HIP kernels
// (vrm_kernels.hip).  Internal; the public contract is include/vrm.h.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

#define VRM_MAX_STEP_AMOUNT 5000 /* raymarcher.wgsl:96 */
#define VRM_STEP_SIZE 0.001f     /* raymarcher.wgsl:126 */
#define VRM_QUEUE_SUBS 4         /* counters per XCD list of the march kernel's tile queues */
#define VRM_QUEUE_BYTES (8 * VRM_QUEUE_SUBS * 128) /* the counters, 128 bytes apart */
#define VRM_OCC_LEVELS 4         /* levels of the occupancy max pyramid (cells of 4, 8, 16, 32 texels) */

// One entry per view of a batch, resident in device memory (read through the scalar cache).
struct VrmViewDesc {
    float ivp[16];       // CameraUniform.inverse_view_proj, column-major (camera.rs:48)
    float pos[4];        // CameraUniform.position
    uint32_t *out;       // RGBA8 target, row pitch = VrmMarchParams.pitch_px words
    float *diag_rgba;    // optional W*H*4 f32 (VRM_OPT_DIAGNOSTICS)
    uint32_t *diag_iters; // optional W*H u32
    float *diag_geom;     // optional W*H*8 f32: rd.xyz, t_min, t_max, #samples inside the box, hit, 0
    float *partial_rgba;  // optional W*H*4 f32: the premultiplied colour and alpha before the 8-bit store (sort-last partials)
};

// Volume as the kernels see it.
struct VrmGridDev {
    // RGBA8 words (r | g<<8 | b<<16 | a<<24), x fastest: idx = x + X*(y + Y*z)  (grid.rs:211).  nullptr (and `bricks` too) while a
    // SCALAR volume is held in its compact form -- packed alpha bricks + occupancy pyramid, 1 B/voxel -- which is all the packed
    // 1-channel march reads; the host re-expands before anything else needs them (vrm_api.cpp grid_expand)
    const uint32_t *vox;
    uint32_t dim[3];
    uint32_t scalar;     // 1 => every voxel has r == g == b == a (DAT volumes, dat.rs:74)
    const uint8_t *occ;  // max alpha per 4x4x4-texel cell, x fastest (exact empty-space skipping): level 0 of a max pyramid
    uint32_t occ_dim[3]; // ceil(dim / 4)
    // level L (1 .. VRM_OCC_LEVELS-1) holds the max over 2^L x 2^L x 2^L level-0 cells, dims ceil(occ_dim / 2^L), at byte
    // offset occ_off[L] of `occ`: the emptiness test of a chunk 2^L times as deep looks at as many cells as a plain one
    uint32_t occ_off[4];
    // the same voxels in bricks of 4(x) x 4(y) x 2(z) texels = 32 words = one 128-byte line:
    // word = 32*(bx + nbx*(by + nby*bz)) + 16*(z&1) + 4*(y&3) + (x&3), bx = x>>2, by = y>>2, bz = z>>1.
    // A staged box of ~12^3 texels touches ~90 lines of it instead of ~190 of the x-fastest array.
    const uint32_t *bricks;
    uint32_t nbx, nby;
    // the alpha channel alone, four x-adjacent texels per word (byte k = texel 4*xw + k), bricked the same way over
    // the packed words: brick = 4 words (16 texels of x) x 4(y) x 2(z) = one 128-byte line, nbxa = ceil(X / 16)
    // bricks along x, nby along y; texels past the volume's edge repeat the edge (clamp-to-edge).  Staging source of
    // the 1-channel march (scalar volumes, transfer function): a quarter of the bytes of `bricks`.
    const uint32_t *abricks;
    uint32_t nbxa;
    float box_min[3];    // VoxelGridUniform.box_min (grid.rs:271)
    float box_max[3];    // box_min + box_size, rounded as the shader computes it (wgsl:202,250)
    float box_size[3];
    float tex_scale[3];  // dim / box_size
    float tex_bias[3];   // -box_min*tex_scale - 0.5
    // Slab partition (sort-last).  box_*, tex_scale are always those of the WHOLE volume; `vox`,
    // `bricks`, `occ` and dim[2] describe the slices this context stores (its slab plus one ghost slice either side),
    // tex_bias[2] is shifted so that texel coordinates are relative to the first stored slice.  A sample belongs to
    // this context iff it passes the in-box test with z in [in_min_z, in_max_z) -- the slab's world-space bounds,
    // box_min[2] / box_max[2] for an unpartitioned volume.  u_lo/u_hi: the texel-coordinate range [u_lo, u_hi) such
    // samples can have (-0.5 and dim - 0.5 unpartitioned).
    float in_min_z, in_max_z;
    float u_lo[3], u_hi[3];
    uint32_t slab;       // 1 => this context owns only part of the z range
};

struct VrmMarchParams {
    VrmGridDev grid;
    const VrmViewDesc *views;
    uint32_t width, height, pitch_px;
    uint32_t tiles_x, tiles_y;     // 16x16-pixel workgroup tiles (variant 0)
    uint32_t sblocks_x, sblocks_y; // 64x64-pixel super-blocks of 8x8-pixel wave tiles (variant 1); x padded to a multiple of 8
    float attenuation;   // voxel_grid.buffer[0] (wgsl:150)
    float col_a[3], col_b[3], col_c[3]; // transfer function colours (grid.rs:285-287)
    uint32_t use_tf;
    uint32_t chunk_steps; // variant 1+: march steps per staged box; 0 = let the launcher choose (about 8 texels of depth)
    // tile queues of the march kernel (nullptr: one workgroup per tile and view): 8 x VRM_QUEUE_SUBS counters, 128 bytes apart, zeroed by the
    // launcher; n_views = the views of this launch
    uint32_t *queue;
    uint32_t n_views;
    uint32_t tiles_total(uint32_t views) const { return sblocks_x * sblocks_y * 64u * views; }
    uint32_t any_far;     // some view of the batch may be a far-camera view (|camera in texels| > 4096): launch the FAR instance too
    int32_t packed_alpha; // 1-channel march staged from the packed alpha bricks (default) or, 0, from the RGBA8 bricks
    // rays per lane of the packed scalar march: 2 = wave tiles of 16x8 pixels, a lane marches two x-adjacent pixels through one
    // staged box per chunk (the host chooses it where a pixel is narrower than ~0.6 texel in every view of the batch); else 1
    uint32_t rays;
    // diagnostics build only (else nullptr): 8 counters the waves add to -- filtered samples, of those with alpha exactly 0,
    // chunk prologues, chunks staged, chunks passed over as empty, deep empty-space probes that found something, chunks
    // marched with global gathers (vrm_diag_read_counters)
    unsigned long long *diag_counters;
};

// launchers implemented in vrm_kernels.hip
hipError_t vrm_launch_march(const VrmMarchParams &p, uint32_t n_views, int variant, bool diagnostics, bool skip_empty,
                            hipStream_t stream);
// bytes of the occupancy pyramid of a dim[0] x dim[1] x dim[2] volume; off[L] = byte offset of level L
size_t vrm_occupancy_bytes(const uint32_t dim[3], uint32_t off[VRM_OCC_LEVELS]);
// `src` and `bps` of the two builders below: the x-fastest RGBA8 words (bps 0), or the raw samples of a scalar volume (bps 1:
// UCHAR, 2: USHORT big-endian) quantised on the fly as dat.rs does -- the direct ingest that never holds the RGBA form
hipError_t vrm_launch_build_occupancy(const void *src, uint32_t bps, const uint32_t dim[3], uint8_t *occ, hipStream_t stream);
hipError_t vrm_launch_build_bricks(const uint32_t *vox, const uint32_t dim[3], uint32_t *bricks, hipStream_t stream);
// bytes of the packed alpha bricks of a dim[0] x dim[1] x dim[2] volume, and their builder
size_t vrm_alpha_bricks_bytes(const uint32_t dim[3]);
hipError_t vrm_launch_build_alpha_bricks(const void *src, uint32_t bps, const uint32_t dim[3], uint32_t *abricks, hipStream_t stream);
// a scalar volume's x-fastest RGBA8 voxels ([b,b,b,b]) back from its packed alpha bricks
hipError_t vrm_launch_expand_alpha_bricks(const uint32_t *abricks, const uint32_t dim[3], uint32_t *vox, hipStream_t stream);
hipError_t vrm_launch_count_samples(const VrmMarchParams &p, uint32_t n_views, unsigned long long *total_dev, hipStream_t stream);
hipError_t vrm_launch_quantize_f32(const float *src_czyx, uint32_t channels, size_t n_voxels, uint32_t *dst, hipStream_t stream);
hipError_t vrm_launch_quantize_scalar(const uint8_t *raw, uint32_t bytes_per_sample, size_t n_voxels, uint32_t *dst,
                                      hipStream_t stream);
hipError_t vrm_launch_fill(uint32_t *dst, size_t n_words, uint32_t value, hipStream_t stream);
// sets *flag_dev to 1 if some voxel has r,g,b not all equal to a (i.e. the grid is not "scalar")
hipError_t vrm_launch_detect_nonscalar(const uint32_t *vox, size_t n_voxels, uint32_t *flag_dev, hipStream_t stream);
// evaluation reductions: `blocks` = vrm_reduce_blocks(n); `partial` holds one value per block, summed by the caller
uint32_t vrm_reduce_blocks(size_t n_items);
hipError_t vrm_launch_compare_volume(const uint32_t *vox, const float *data_czyx, size_t n_voxels, bool density_only,
                                     const float tf9[9], double *partial, uint32_t blocks, hipStream_t stream);
hipError_t vrm_launch_ssd_u8(const uint8_t *a, const uint8_t *b, size_t row_words, size_t rows, size_t pitch_a, size_t pitch_b,
                             unsigned long long *partial, uint32_t blocks, hipStream_t stream);
// a, b: width*height*channels interleaved bytes, width and height >= 7; items = (width-6)*(height-6)*channels
hipError_t vrm_launch_ssim_u8(const uint8_t *a, const uint8_t *b, uint32_t width, uint32_t height, uint32_t channels, double c1,
                              double c2, double *partial, uint32_t blocks, hipStream_t stream);

// sort-last compositing: the part of a view the compositing kernel needs (the sign of rd.z per pixel)
struct VrmCompositeView {
    float ivp[16];
};
hipError_t vrm_launch_composite(const float *partials, uint32_t n_slabs, const VrmCompositeView &cv, uint32_t width, uint32_t height,
                                uint32_t row0, uint32_t rows, uint32_t *out_rgba8, float *out_f32, hipStream_t stream);
