// Replacement for src/voxel/grid.rs of bennetde/volume-renderer (reference grid.rs:9-299).
// UNVERIFIED: never compiled (no Rust toolchain in the image this was written in).
//
// VoxelGrid keeps every member the rest of the reference touches, with the reference's signatures:
//     VoxelGrid::new(dimensions, &device, &queue)                 grid.rs:44   (mod.rs:29, dat.rs:57, netcdf.rs:47)
//     set_color / Index<UVec3> / IndexMut<UVec3>                  grid.rs:202, 249-263 (loaders, compare/netcdf.rs:24)
//     update_buffer(&queue)                                       grid.rs:214  (mod.rs:34, dat.rs:80, netcdf.rs:66)
//     update_voxel_grid_buffer(&queue)                            grid.rs:240  (state.rs:367)
//     update_transfer_function_buffer(&queue)                     grid.rs:244  (state.rs:373)
//     pub dimensions, pub attenuation, pub transfer_function_colors               (state.rs:296, 365, 370-372; netcdf.rs:11-13)
//     VoxelGridUniform, RaymarchTransferFunctionColors (+ use_transfer_function_active / set_transfer_function_active)
// so loaders/dat.rs, loaders/netcdf.rs, compare/netcdf.rs and state.rs compile unchanged.  The members that only
// ray_marcher/mod.rs used -- the two bind-group layouts, the bind groups, the Texture3D -- are gone with the wgpu pipeline
// (grep: mod.rs:60-61, 135-136 are their only users).  The voxels stay in the host Vec exactly as in the reference (the
// loaders fill it voxel by voxel, compare/netcdf.rs reads it back); `update_buffer` hands them to the library in one call,
// where the reference did one queue.write_texture.
use std::ops::{Index, IndexMut};

use glam::{UVec3, Vec3Swizzles};
use wgpu::{Device, Queue};

use super::voxel::Voxel;
use crate::ray_marcher::{check, current_ctx};

pub struct VoxelGrid {
    voxels: Vec<Voxel>,
    pub dimensions: UVec3,
    pub attenuation: f32,
    pub transfer_function_colors: RaymarchTransferFunctionColors,
}

#[repr(C)]
#[derive(Debug, Copy, Clone, bytemuck::Pod, bytemuck::Zeroable)]
pub struct VoxelGridUniform {
    dimensions: [u32; 4],
    box_min: [f32; 4],
    box_size: [f32; 4],
    buffer: [f32; 4],
}

#[repr(C)]
#[derive(Debug, Copy, Clone, bytemuck::Pod, bytemuck::Zeroable)]
pub struct RaymarchTransferFunctionColors {
    pub color_a: [f32; 4],
    pub color_b: [f32; 4],
    pub color_c: [f32; 4],
    pub use_transfer_function: [u32; 4],
}

impl VoxelGrid {
    /// grid.rs:44-200: a fresh grid of Voxel::default() = [0,0,0,255]; attenuation 1.0, default transfer function
    /// (grid.rs:197-198).  `device` and `queue` are accepted and not used; the library context is the thread's RayMarcher's.
    pub fn new(dimensions: UVec3, _device: &Device, _queue: &Queue) -> Self {
        check(unsafe { vrm_sys::vrm_grid_resize(current_ctx(), dimensions.x, dimensions.y, dimensions.z) }); // resets attenuation and TF too
        Self {
            voxels: vec![Voxel::default(); (dimensions.x * dimensions.y * dimensions.z) as usize],   // grid.rs:83
            dimensions,
            attenuation: 1.0,
            transfer_function_colors: RaymarchTransferFunctionColors::new(),
        }
    }

    pub fn set_color(&mut self, position: UVec3, color: [u8; 4]) {
        let index = self.get_index(position);
        self.voxels[index].set_color(color);
    }

    fn get_index(&self, position: UVec3) -> usize {
        if position.x >= self.dimensions.x || position.y >= self.dimensions.y || position.z >= self.dimensions.z {
            panic!("Tried to access grid outside array")   // grid.rs:209
        }
        (position.x + self.dimensions.x * (position.y + self.dimensions.y * position.z)) as usize      // grid.rs:211
    }

    /// grid.rs:214-238: the whole Vec<Voxel> (x fastest, 4 bytes per voxel) in one upload.
    pub fn update_buffer(&self, _queue: &Queue) {
        let bytes: &[u8] = bytemuck::cast_slice(&self.voxels);
        check(unsafe { vrm_sys::vrm_grid_upload_rgba8(current_ctx(), bytes.as_ptr(), bytes.len()) });
    }

    /// grid.rs:240-242 (GUI attenuation slider, state.rs:365-368)
    pub fn update_voxel_grid_buffer(&mut self, _queue: &Queue) {
        check(unsafe { vrm_sys::vrm_set_attenuation(current_ctx(), self.attenuation) });
    }

    /// grid.rs:244-246 (GUI transfer-function checkbox, state.rs:370-374)
    pub fn update_transfer_function_buffer(&mut self, _queue: &Queue) {
        let tf = &self.transfer_function_colors as *const RaymarchTransferFunctionColors as *const vrm_sys::vrm_transfer_function;
        check(unsafe { vrm_sys::vrm_set_transfer_function(current_ctx(), tf) });
    }
}

impl Index<UVec3> for VoxelGrid {
    type Output = Voxel;
    fn index(&self, index: UVec3) -> &Self::Output {
        let index = self.get_index(index);
        &self.voxels[index]
    }
}

impl IndexMut<UVec3> for VoxelGrid {
    fn index_mut(&mut self, index: UVec3) -> &mut Self::Output {
        let index = self.get_index(index);
        &mut self.voxels[index]
    }
}

impl VoxelGridUniform {
    /// grid.rs:265-280 (kept for parity of the type; the library computes the same values, vrm_get_grid_uniform)
    pub fn new(dimensions: UVec3, attenuation: f32) -> Self {
        let min_dimension = u32::min(dimensions.x, u32::min(dimensions.y, dimensions.z));
        let box_size = dimensions.as_vec3() / min_dimension as f32;
        let box_min = -box_size / 2.0;
        Self {
            dimensions: dimensions.xyzx().to_array(),
            box_min: [box_min.x, box_min.y, box_min.z, 0.0],
            box_size: [box_size.x, box_size.y, box_size.z, 0.0],
            buffer: [attenuation; 4],
        }
    }
}

impl RaymarchTransferFunctionColors {
    pub fn new() -> Self {   // grid.rs:282-290
        Self { color_a: [0.117, 0.188, 0.62, 1.0], color_b: [0.7294, 0.7294, 0.7294, 1.0], color_c: [0.5725, 0.0, 0.039, 1.0], use_transfer_function: [0; 4] }
    }
    pub fn use_transfer_function_active(&self) -> bool { self.use_transfer_function[0] != 0 }
    pub fn set_transfer_function_active(&mut self, val: bool) { self.use_transfer_function[0] = val as u32; }
}
