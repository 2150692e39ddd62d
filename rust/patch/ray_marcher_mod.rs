// Replacement for /src/ray_marcher/mod.rs of bennetde/volume-renderer (reference mod.rs:18-142).
// UNVERIFIED: never compiled (no Rust toolchain in the image this was written in).
//
// RayMarcher keeps its public shape -- `new`, `draw`, `pub voxel_grid` -- while the wgpu pipeline, the bind groups and
// the fullscreen triangle go away: the march runs in libvrm.so on the MI355X.  `VoxelGrid` (voxel/grid.rs) keeps its
// fields and forwards its three `update_*` methods (see INTEGRATION.md section 2 for the call-site table).
use std::ffi::CStr;

use glam::UVec3;
use wgpu::SurfaceConfiguration;

use crate::camera::CameraUniform;
use crate::voxel::grid::VoxelGrid;

pub struct RayMarcher {
    ctx: *mut vrm_sys::vrm_ctx,
    target: *mut vrm_sys::vrm_image,
    pub voxel_grid: VoxelGrid,
}

fn check(status: std::os::raw::c_int) {
    if status != vrm_sys::VRM_OK {
        let msg = unsafe { CStr::from_ptr(vrm_sys::vrm_last_error()) }.to_string_lossy().into_owned();
        panic!("vrm status {status}: {msg}"); // the reference unwrap()s/panics at the same places
    }
}

impl RayMarcher {
    // was: RayMarcher::new(&Device, &Queue, &SurfaceConfiguration, Rc<BindGroup>)   (mod.rs:26)
    pub fn new(config: &SurfaceConfiguration) -> Self {
        let cfg = vrm_sys::vrm_config { device_id: 0, width: config.width, height: config.height };
        let (mut ctx, mut img) = (std::ptr::null_mut(), std::ptr::null_mut());
        check(unsafe { vrm_sys::vrm_create(&cfg, &mut ctx) }); // starts with the 3x3x3 grid of [0,0,0,255] (mod.rs:29)
        check(unsafe { vrm_sys::vrm_image_create(ctx, config.width, config.height, &mut img) });
        RayMarcher { ctx, target: img, voxel_grid: VoxelGrid::attach(ctx, UVec3::new(3, 3, 3)) }
    }

    // was: draw(&self, &Device, &TextureView) -> CommandBuffer   (mod.rs:113): enqueue only, like the unsubmitted
    // CommandBuffer; `camera_uniform` is the struct State already fills every frame (state.rs:237)
    pub fn draw(&self, camera_uniform: &CameraUniform) {
        check(unsafe { vrm_sys::vrm_draw(self.ctx, camera_uniform as *const CameraUniform as *const vrm_sys::vrm_camera_uniform, self.target) });
    }

    // Screenshotter::screenshot + save_screenshot_to_disk (screenshot.rs:30-92) on the render target
    pub fn save_png(&self, path: &str) {
        let c = std::ffi::CString::new(path).unwrap();
        check(unsafe { vrm_sys::vrm_image_readback_async(self.target) });
        check(unsafe { vrm_sys::vrm_image_save_png(self.target, c.as_ptr()) });
    }

    pub fn raw(&self) -> *mut vrm_sys::vrm_ctx { self.ctx }
}

impl Drop for RayMarcher {
    fn drop(&mut self) {
        unsafe {
            vrm_sys::vrm_image_destroy(self.target);
            vrm_sys::vrm_destroy(self.ctx);
        }
    }
}
