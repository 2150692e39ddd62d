//! Raw declarations of every entry point of `include/vrm.h`, one to one.
//!
//! UNVERIFIED: the image this was written in has no `cargo`/`rustc`; the file has never been
//! compiled.  `tests/test_abi_and_host.py` checks that it names every symbol the library exports and the header
//! declares, nothing more.  Struct layouts follow the `#[repr(C)]` uniforms of the reference:
//! `CameraUniform` (camera.rs:42-50), `VoxelGridUniform` (grid.rs:24-32), `RaymarchTransferFunctionColors`
//! (grid.rs:34-41).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int};

#[repr(C)] pub struct vrm_ctx { _p: [u8; 0] }
#[repr(C)] pub struct vrm_image { _p: [u8; 0] }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct vrm_config { pub device_id: i32, pub width: u32, pub height: u32 }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct vrm_camera_uniform { pub position: [f32; 4], pub view_proj: [f32; 16], pub inverse_view_proj: [f32; 16] }

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct vrm_transfer_function {
    pub color_a: [f32; 4], pub color_b: [f32; 4], pub color_c: [f32; 4], pub use_transfer_function: [u32; 4],
}

#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct vrm_grid_uniform { pub dimensions: [u32; 4], pub box_min: [f32; 4], pub box_size: [f32; 4], pub buffer: [f32; 4] }

/// an ncclUniqueId: made on rank 0 (vrm_comm_unique_id), carried to the other ranks by the caller's own means
#[repr(C)] #[derive(Clone, Copy)]
pub struct vrm_comm_id { pub bytes: [u8; 128] }

pub const VRM_OK: c_int = 0;
pub const VRM_ERR_INVALID: c_int = -1;
pub const VRM_ERR_NO_DEVICE: c_int = -2;
pub const VRM_ERR_HIP: c_int = -3;
pub const VRM_ERR_IO: c_int = -4;
pub const VRM_ERR_STATE: c_int = -5;

pub const VRM_OPT_KERNEL: c_int = 0;
pub const VRM_OPT_SKIP_EMPTY: c_int = 1;
pub const VRM_OPT_PROFILE: c_int = 2;
pub const VRM_OPT_DIAGNOSTICS: c_int = 3;
pub const VRM_OPT_CHUNK_STEPS: c_int = 4;
pub const VRM_OPT_PACKED_ALPHA: c_int = 5;
pub const VRM_OPT_COMPACT_SCALAR: c_int = 6;
pub const VRM_OPT_RAYS_PER_LANE: c_int = 7;

extern "C" {
    // RayMarcher (ray_marcher/mod.rs:18-142)
    pub fn vrm_create(config: *const vrm_config, out_ctx: *mut *mut vrm_ctx) -> c_int;
    pub fn vrm_destroy(ctx: *mut vrm_ctx);
    pub fn vrm_draw(ctx: *mut vrm_ctx, camera: *const vrm_camera_uniform, target: *mut vrm_image) -> c_int;
    pub fn vrm_draw_batch(ctx: *mut vrm_ctx, cameras: *const vrm_camera_uniform, n: u32, targets: *const *mut vrm_image) -> c_int;
    // the same draw with the kernel's descriptor scratch in a static device array: allocates nothing, n <= 256 (no reference counterpart)
    pub fn vrm_draw_batch_static(ctx: *mut vrm_ctx, cameras: *const vrm_camera_uniform, n: u32, targets: *const *mut vrm_image) -> c_int;
    pub fn vrm_synchronize(ctx: *mut vrm_ctx) -> c_int;
    // VoxelGrid (voxel/grid.rs:9-299) and the loaders' rules (loaders/dat.rs, loaders/netcdf.rs)
    pub fn vrm_grid_resize(ctx: *mut vrm_ctx, x: u32, y: u32, z: u32) -> c_int;
    pub fn vrm_grid_upload_rgba8(ctx: *mut vrm_ctx, host_voxels: *const u8, nbytes: usize) -> c_int;
    pub fn vrm_grid_upload_rgba8_device(ctx: *mut vrm_ctx, device_voxels: *const u8, nbytes: usize) -> c_int;
    pub fn vrm_grid_upload_f32(ctx: *mut vrm_ctx, host_czyx: *const f32, channels: u32) -> c_int;
    pub fn vrm_grid_upload_f32_device(ctx: *mut vrm_ctx, device_czyx: *const f32, channels: u32) -> c_int;
    pub fn vrm_grid_upload_scalar(ctx: *mut vrm_ctx, host_raw: *const u8, nbytes: usize, bytes_per_sample: u32) -> c_int;
    pub fn vrm_grid_download_rgba8(ctx: *mut vrm_ctx, host_voxels: *mut u8, nbytes: usize) -> c_int;
    pub fn vrm_set_attenuation(ctx: *mut vrm_ctx, attenuation: f32) -> c_int;
    pub fn vrm_set_transfer_function(ctx: *mut vrm_ctx, tf: *const vrm_transfer_function) -> c_int;
    pub fn vrm_get_grid_uniform(ctx: *mut vrm_ctx, out: *mut vrm_grid_uniform) -> c_int;
    // Screenshotter (screenshot.rs:6-97)
    pub fn vrm_image_create(ctx: *mut vrm_ctx, width: u32, height: u32, out_image: *mut *mut vrm_image) -> c_int;
    pub fn vrm_image_destroy(image: *mut vrm_image);
    pub fn vrm_image_readback_async(image: *mut vrm_image) -> c_int;
    pub fn vrm_image_map(image: *mut vrm_image, rgba8: *mut *const u8, pitch: *mut usize) -> c_int;
    pub fn vrm_image_unmap(image: *mut vrm_image) -> c_int;
    pub fn vrm_image_save_png(image: *mut vrm_image, path: *const c_char) -> c_int;
    // camera, view list, export loop, DAT loader
    pub fn vrm_camera_look_at(position: *const f32, target: *const f32, aspect: f32, out: *mut vrm_camera_uniform, basis9: *mut f32) -> c_int;
    pub fn vrm_sphere_position(ix: u32, iy: u32, x_div: u32, y_div: u32, radius: f32, origin: *const f32, out_position: *mut f32) -> c_int;
    pub fn vrm_export_sphere(ctx: *mut vrm_ctx, x_div: u32, y_div: u32, radius: f32, origin: *const f32, out_dir: *const c_char,
                             part: u32, n_parts: u32) -> c_int;
    pub fn vrm_load_dat(ctx: *mut vrm_ctx, dat_path: *const c_char) -> c_int;
    // NetCDF interchange (loaders/netcdf.rs)
    pub fn vrm_load_netcdf(ctx: *mut vrm_ctx, path: *const c_char) -> c_int;
    pub fn vrm_write_netcdf(ctx: *mut vrm_ctx, path: *const c_char) -> c_int;
    pub fn vrm_write_netcdf4(ctx: *mut vrm_ctx, path: *const c_char) -> c_int;
    pub fn vrm_netcdf_read_color(path: *const c_char, x: *mut u32, y: *mut u32, z: *mut u32, color_czyx: *mut f32, capacity_floats: usize) -> c_int;
    pub fn vrm_netcdf_write_color(path: *const c_char, x: u32, y: u32, z: u32, color_czyx: *const f32) -> c_int;
    pub fn vrm_netcdf_write_color_nc4(path: *const c_char, x: u32, y: u32, z: u32, color_czyx: *const f32) -> c_int;
    // sort-last rendering of a z-slab-partitioned volume
    pub fn vrm_grid_resize_slab(ctx: *mut vrm_ctx, x: u32, y: u32, z: u32, z_begin: u32, z_end: u32) -> c_int;
    pub fn vrm_grid_resident_bytes(ctx: *mut vrm_ctx, bytes: *mut usize) -> c_int;
    pub fn vrm_grid_get_slab(ctx: *mut vrm_ctx, z_store_begin: *mut u32, z_store_end: *mut u32, z_begin: *mut u32, z_end: *mut u32) -> c_int;
    pub fn vrm_draw_partial(ctx: *mut vrm_ctx, cameras: *const vrm_camera_uniform, n: u32, width: u32, height: u32,
                            rgba_f32_device: *mut f32) -> c_int;
    pub fn vrm_composite(ctx: *mut vrm_ctx, camera: *const vrm_camera_uniform, partials_device: *const f32, n_slabs: u32, width: u32,
                         height: u32, row0: u32, rows: u32, out_rgba8_device: *mut u8, out_f32_device: *mut f32) -> c_int;
    // evaluation (compare/netcdf.rs, scripts/similarity.py)
    pub fn vrm_compare_volume_f32(ctx: *mut vrm_ctx, color_czyx: *const f32, density_only: c_int, out_rmse: *mut f32) -> c_int;
    pub fn vrm_compare_volume_f32_device(ctx: *mut vrm_ctx, color_czyx_device: *const f32, density_only: c_int, out_rmse: *mut f32) -> c_int;
    pub fn vrm_image_ssd(a: *mut vrm_image, b: *mut vrm_image, out_ssd: *mut u64, out_count: *mut u64) -> c_int;
    pub fn vrm_ssd_u8(ctx: *mut vrm_ctx, a: *const u8, b: *const u8, n: usize, out_ssd: *mut u64) -> c_int;
    pub fn vrm_ssim_u8(ctx: *mut vrm_ctx, a: *const u8, b: *const u8, width: u32, height: u32, channels: u32, data_range: f64,
                       out_ssim: *mut f64) -> c_int;
    pub fn vrm_png_read(path: *const c_char, width: *mut u32, height: *mut u32, channels: *mut u32, pixels: *mut u8, capacity: usize) -> c_int;
    // build extensions
    pub fn vrm_set_option(ctx: *mut vrm_ctx, option: c_int, value: i64) -> c_int;
    pub fn vrm_get_option(ctx: *mut vrm_ctx, option: c_int, value: *mut i64) -> c_int;
    pub fn vrm_profile_read(ctx: *mut vrm_ctx, kernel_ms: *mut f64, launches: *mut u64, reset: c_int) -> c_int;
    pub fn vrm_image_read_diagnostics(image: *mut vrm_image, rgba_f32: *mut f32, iterations: *mut u32) -> c_int;
    pub fn vrm_image_read_sample_counts(image: *mut vrm_image, executed: *mut u32) -> c_int;
    pub fn vrm_image_read_ray_diagnostics(image: *mut vrm_image, geom8: *mut f32) -> c_int;
    pub fn vrm_diag_read_counters(ctx: *mut vrm_ctx, out: *mut u64, reset: c_int) -> c_int;
    pub fn vrm_count_reference_samples(ctx: *mut vrm_ctx, cameras: *const vrm_camera_uniform, n: u32, width: u32, height: u32,
                                       out_total: *mut u64) -> c_int;
    // the volume broadcast of the view-partitioned export inside the library (RCCL, bound at first use); no reference counterpart
    pub fn vrm_comm_unique_id(id: *mut vrm_comm_id) -> c_int;
    pub fn vrm_comm_init(ctx: *mut vrm_ctx, id: *const vrm_comm_id, rank: c_int, n_ranks: c_int) -> c_int;
    pub fn vrm_grid_broadcast(ctx: *mut vrm_ctx, root: c_int) -> c_int;
    pub fn vrm_comm_destroy(ctx: *mut vrm_ctx) -> c_int;
    pub fn vrm_last_error() -> *const c_char;
    pub fn vrm_version() -> *const c_char;
}
