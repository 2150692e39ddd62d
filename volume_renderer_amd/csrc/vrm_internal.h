// vrm_internal.h -- host-side objects behind the opaque handles of include/vrm.h.
#pragma once
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "../../include/vrm.h"
#include "vrm_device.h"

struct vrm_ctx {
    int device = 0;
    uint32_t def_width = 0, def_height = 0;
    hipStream_t stream = nullptr;      // every kernel of this context is launched here
    hipStream_t copy_stream = nullptr; // device->host readbacks, ordered behind draws by events

    // VoxelGrid (voxel/grid.rs:9-22)
    uint32_t dim[3] = {0, 0, 0};  // of the stored voxels
    // slab partition (sort-last): the whole volume is gdim; this context stores z slices [z_store0, z_store0 + dim[2])
    // and owns the samples of slices [slab_z0, slab_z1).  Unpartitioned: gdim == dim, z_store0 = 0, slab = [0, dim[2]).
    uint32_t gdim[3] = {0, 0, 0};
    uint32_t z_store0 = 0, slab_z0 = 0, slab_z1 = 0;
    uint32_t *vox = nullptr; // RGBA8 words, x fastest (nullptr, like `bricks`, while a scalar volume is held compact: vrm_api.cpp grid_compact)
    size_t vox_words = 0;
    bool scalar = false;     // r == g == b == a everywhere
    // the grid was resized and nothing has been uploaded yet: its contents are Voxel::default by promise -- `vox` and `bricks`
    // are not allocated, `abricks` and `occ` not filled -- until something looks at it (vrm_api.cpp grid_expand)
    bool lazy_default = false;
    uint8_t *occ = nullptr;  // max alpha per 4x4x4-texel cell, rebuilt whenever the voxels change
    uint32_t *bricks = nullptr; // the same voxels in 4x4x2-texel bricks of one 128-byte line each (staging source)
    uint32_t *abricks = nullptr; // the alpha channel alone, 4 texels per word, bricked (1-channel march staging source)
    float attenuation = 1.0f;
    vrm_transfer_function tf;
    uint32_t *flag_dev = nullptr; // 1 word scratch for detect_nonscalar
    uint32_t *queue_dev = nullptr; // 1 KB: the march kernel's tile-queue counters (8, 128 bytes apart)

    // batch descriptors (device + pinned staging)
    VrmViewDesc *views_dev = nullptr;
    VrmViewDesc *views_host = nullptr; // pinned
    uint32_t views_cap = 0;
    void *scratch = nullptr;           // this device's address of the static descriptor scratch (vrm_draw_batch_static)
    hipEvent_t views_free = nullptr;   // signalled when the last batch's descriptor upload was consumed

    // render targets created from this context and not yet destroyed: vrm_destroy orphans them (ctx = nullptr) so that a
    // later vrm_image_destroy frees their memory without touching the context's streams
    std::vector<struct vrm_image *> images;

    // RCCL communicator of vrm_comm_init (an ncclComm_t; nullptr = none): the volume broadcast of the view-partitioned export
    void *comm = nullptr;
    int comm_rank = 0, comm_size = 0;

    // options
    int64_t opt_kernel = 1, opt_skip = 1, opt_profile = 0, opt_diag = 0, opt_chunk = 0, opt_packed = 1, opt_compact = 1, opt_rays = 0;

    // profiling (VRM_OPT_PROFILE)
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    static constexpr uint32_t kBatchEvents = 64; // ring of per-launch "drawn" events, created with the context
    hipEvent_t batch_events[kBatchEvents] = {};
    uint32_t batch_next = 0;
    size_t prof_used = 0;
    double prof_ms = 0.0;
    uint64_t prof_launches = 0;
};

// (vrm_api.cpp) a scalar volume held in its compact form gets its x-fastest RGBA8 copy and RGBA bricks back; no-op otherwise
int vrm_grid_expand_internal(vrm_ctx *ctx);

struct vrm_image {
    vrm_ctx *ctx = nullptr;      // nullptr once the context was destroyed: only vrm_image_destroy is valid then
    int device = 0;
    uint32_t width = 0, height = 0;
    size_t pitch = 0;            // bytes, ceil(4w/256)*256 (screenshot.rs:94-96)
    uint32_t *dev = nullptr;     // render target
    uint8_t *host = nullptr;     // pinned readback buffer (the MAP_READ buffer of screenshot.rs:16-22)
    float *diag_rgba = nullptr;  // lazily allocated with VRM_OPT_DIAGNOSTICS
    uint32_t *diag_iters = nullptr;
    float *diag_geom = nullptr;
    hipEvent_t drawn = nullptr;  // recorded on ctx->stream when the image was created (its clear)
    // the event behind the last draw into this image: ONE per launch, shared by the launch's targets, from the context's ring
    // (an event per target and launch was 64 markers between two kernels: 0.3 ms of idle chip per launch).  A ring slot
    // that has been re-recorded since stands for a LATER point of the same in-order stream: waiting on it is still enough.
    hipEvent_t drawn_batch = nullptr;
    hipEvent_t copied = nullptr; // recorded on ctx->copy_stream after the last readback
    bool has_draw = false, copy_pending = false, copy_valid = false, mapped = false;
};

void vrm_set_error(const char *fmt, ...);
int vrm_check_hip(hipError_t e, const char *what);

// "Nothing aborts across the boundary" (include/vrm.h): an entry point whose body can throw (std::vector, std::string,
// std::thread) runs it through this, so that an allocation failure or a length error comes back as a status.
template <class F>
static inline int vrm_guard(const char *entry, int code, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        vrm_set_error("%s: out of host memory", entry);
    } catch (const std::exception &e) {
        vrm_set_error("%s: %s", entry, e.what());
    } catch (...) {
        vrm_set_error("%s: unknown exception", entry);
    }
    return code;
}

// host helpers implemented in vrm_host.cpp
int vrm_write_png_rgba8(const char *path, const uint8_t *rgba, uint32_t w, uint32_t h, size_t pitch);
