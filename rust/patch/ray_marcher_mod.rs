// Replacement for src/ray_marcher/mod.rs of bennetde/volume-renderer (reference mod.rs:18-142).
// UNVERIFIED: never compiled (no Rust toolchain in the image this was written in).
//
// RayMarcher keeps the reference's public interface EXACTLY:
//     pub fn new(device: &Device, queue: &Queue, config: &SurfaceConfiguration, camera_bind_group: Rc<BindGroup>) -> Self   (mod.rs:26)
//     pub fn draw(&self, device: &Device, view: &TextureView) -> CommandBuffer                                              (mod.rs:113)
//     pub voxel_grid: VoxelGrid                                                                                               (mod.rs:22)
// so src/state.rs compiles unchanged (state.rs:154, 246, 273, 281, 288, 297, 365-373).  What is behind it changes: the wgpu
// render pipeline and raymarcher.wgsl go away, the march runs in libvrm.so on the MI355X (vrm_draw), and `draw` returns a
// command buffer that puts the finished image into `view` with a one-triangle blit, so the window, the egui pass drawn on top
// (gui.rs:84-98, LoadOp::Load) and the Screenshotter (which copies the surface texture, screenshot.rs:30-54) see exactly the
// pixels the library produced (Rgba8Unorm, nearest, 1:1).
//
// Where the camera comes from: the reference's shader reads the CameraUniform from `camera_bind_group`, a wgpu buffer that
// State::update fills with queue.write_buffer (state.rs:237-238); a buffer created UNIFORM | COPY_DST (state.rs:114) cannot be
// read back.  The same CameraUniform value is therefore published by camera.rs itself: `CameraUniform::update_view_proj`
// (camera.rs:61-67, called at state.rs:108 and 237, i.e. before every draw) stores a copy in a thread-local that `draw` reads --
// a two-line patch of camera.rs (rust/patch/camera_rs.diff), none of state.rs.  `camera_bind_group` and `queue` are accepted
// and not used.
use std::cell::RefCell;
use std::ffi::CStr;
use std::rc::Rc;

use glam::UVec3;
use wgpu::{BindGroup, Color, CommandBuffer, Device, Queue, SurfaceConfiguration, TextureView};

use crate::camera::CameraUniform;
use crate::voxel::grid::VoxelGrid;

thread_local! {
    /// The library context of this thread's RayMarcher.  VoxelGrid::new (voxel/grid.rs) has no argument to receive it through
    /// -- the loaders build a fresh grid with `*grid = VoxelGrid::new(resolution, &device, &queue)` (dat.rs:57, netcdf.rs:47) --
    /// and the reference is single-threaded with one RayMarcher (Rc<BindGroup>, mod.rs:21), so the context is per thread.
    pub(crate) static VRM_CTX: RefCell<*mut vrm_sys::vrm_ctx> = RefCell::new(std::ptr::null_mut());
    /// Last CameraUniform computed by CameraUniform::update_view_proj (rust/patch/camera_rs.diff).
    pub(crate) static LAST_CAMERA: RefCell<Option<CameraUniform>> = RefCell::new(None);
}

pub(crate) fn check(status: std::os::raw::c_int) {
    if status != vrm_sys::VRM_OK {
        let msg = unsafe { CStr::from_ptr(vrm_sys::vrm_last_error()) }.to_string_lossy().into_owned();
        panic!("vrm status {status}: {msg}"); // the reference unwrap()s / panics at the same places
    }
}

pub(crate) fn current_ctx() -> *mut vrm_sys::vrm_ctx {
    let ctx = VRM_CTX.with(|c| *c.borrow());
    assert!(!ctx.is_null(), "VoxelGrid used before RayMarcher::new");
    ctx
}

const BLIT_WGSL: &str = r#"
@group(0) @binding(0) var img: texture_2d<f32>;
@vertex fn vs(@builtin(vertex_index) i: u32) -> @builtin(position) vec4<f32> {
    let p = array<vec2<f32>, 3>(vec2(-1.0, -1.0), vec2(3.0, -1.0), vec2(-1.0, 3.0));   // the reference's triangle, mod.rs:8-12
    return vec4(p[i], 0.0, 1.0);
}
@fragment fn fs(@builtin(position) pos: vec4<f32>) -> @location(0) vec4<f32> {
    return textureLoad(img, vec2<i32>(pos.xy), 0);                                      // 1:1, no filtering, REPLACE blend
}
"#;

pub struct RayMarcher {
    ctx: *mut vrm_sys::vrm_ctx,
    target: *mut vrm_sys::vrm_image,       // render target + pinned read-back buffer inside the library
    width: u32,
    height: u32,
    staging: wgpu::Buffer,                 // MAP_WRITE | COPY_SRC, rows 256-byte aligned like the library's (screenshot.rs:94-96)
    blit_texture: wgpu::Texture,
    blit_bind_group: BindGroup,
    blit_pipeline: wgpu::RenderPipeline,
    pub voxel_grid: VoxelGrid,
}

impl RayMarcher {
    pub fn new(device: &Device, queue: &Queue, config: &SurfaceConfiguration, _camera_bind_group: Rc<BindGroup>) -> Self {
        let cfg = vrm_sys::vrm_config { device_id: 0, width: config.width, height: config.height };
        let (mut ctx, mut img) = (std::ptr::null_mut(), std::ptr::null_mut());
        check(unsafe { vrm_sys::vrm_create(&cfg, &mut ctx) }); // starts with the 3x3x3 grid of [0,0,0,255] (mod.rs:29)
        check(unsafe { vrm_sys::vrm_image_create(ctx, config.width, config.height, &mut img) });
        VRM_CTX.with(|c| *c.borrow_mut() = ctx);
        let voxel_grid = VoxelGrid::new(UVec3::new(3, 3, 3), device, queue);           // mod.rs:29
        voxel_grid.update_buffer(queue);                                               // mod.rs:34

        let pitch = ((4 * config.width + 255) / 256) * 256;
        let staging = device.create_buffer(&wgpu::BufferDescriptor {
            label: Some("vrm staging"), size: (pitch * config.height) as u64,
            usage: wgpu::BufferUsages::MAP_WRITE | wgpu::BufferUsages::COPY_SRC, mapped_at_creation: false,
        });
        let blit_texture = device.create_texture(&wgpu::TextureDescriptor {
            label: Some("vrm image"), size: wgpu::Extent3d { width: config.width, height: config.height, depth_or_array_layers: 1 },
            mip_level_count: 1, sample_count: 1, dimension: wgpu::TextureDimension::D2, format: wgpu::TextureFormat::Rgba8Unorm,
            usage: wgpu::TextureUsages::COPY_DST | wgpu::TextureUsages::TEXTURE_BINDING, view_formats: &[],
        });
        let layout = device.create_bind_group_layout(&wgpu::BindGroupLayoutDescriptor {
            label: Some("vrm blit layout"),
            entries: &[wgpu::BindGroupLayoutEntry {
                binding: 0, visibility: wgpu::ShaderStages::FRAGMENT, count: None,
                ty: wgpu::BindingType::Texture { sample_type: wgpu::TextureSampleType::Float { filterable: false },
                                                 view_dimension: wgpu::TextureViewDimension::D2, multisampled: false },
            }],
        });
        let blit_bind_group = device.create_bind_group(&wgpu::BindGroupDescriptor {
            label: Some("vrm blit"), layout: &layout,
            entries: &[wgpu::BindGroupEntry { binding: 0,
                resource: wgpu::BindingResource::TextureView(&blit_texture.create_view(&wgpu::TextureViewDescriptor::default())) }],
        });
        let shader = device.create_shader_module(wgpu::ShaderModuleDescriptor { label: Some("vrm blit"), source: wgpu::ShaderSource::Wgsl(BLIT_WGSL.into()) });
        let pl = device.create_pipeline_layout(&wgpu::PipelineLayoutDescriptor { label: None, bind_group_layouts: &[&layout], push_constant_ranges: &[] });
        let blit_pipeline = device.create_render_pipeline(&wgpu::RenderPipelineDescriptor {
            label: Some("vrm blit"), layout: Some(&pl),
            vertex: wgpu::VertexState { module: &shader, entry_point: "vs", buffers: &[] },
            fragment: Some(wgpu::FragmentState { module: &shader, entry_point: "fs",
                targets: &[Some(wgpu::ColorTargetState { format: config.format, blend: Some(wgpu::BlendState::REPLACE), // mod.rs:79-83
                                                        write_mask: wgpu::ColorWrites::ALL })] }),
            primitive: wgpu::PrimitiveState::default(), depth_stencil: None, multisample: wgpu::MultisampleState::default(), multiview: None,
        });
        RayMarcher { ctx, target: img, width: config.width, height: config.height, staging, blit_texture, blit_bind_group, blit_pipeline, voxel_grid }
    }

    pub fn draw(&self, device: &Device, view: &TextureView) -> CommandBuffer {
        // 1. the march, on the MI355X
        let cam = LAST_CAMERA.with(|c| *c.borrow()).expect("CameraUniform::update_view_proj was never called");
        check(unsafe { vrm_sys::vrm_draw(self.ctx, &cam as *const CameraUniform as *const vrm_sys::vrm_camera_uniform, self.target) });
        check(unsafe { vrm_sys::vrm_image_readback_async(self.target) });
        let (mut px, mut pitch) = (std::ptr::null(), 0usize);
        check(unsafe { vrm_sys::vrm_image_map(self.target, &mut px, &mut pitch) });   // blocks until the copy landed
        // 2. into the wgpu staging buffer (same 256-byte row pitch on both sides)
        let slice = self.staging.slice(..);
        slice.map_async(wgpu::MapMode::Write, |r| r.expect("map staging buffer"));
        device.poll(wgpu::Maintain::Wait);                                              // like screenshot.rs:62
        slice.get_mapped_range_mut().copy_from_slice(unsafe { std::slice::from_raw_parts(px, pitch * self.height as usize) });
        self.staging.unmap();
        check(unsafe { vrm_sys::vrm_image_unmap(self.target) });
        // 3. the command buffer State::render submits (state.rs:380-387): staging -> texture, then one triangle into `view`
        let mut encoder = device.create_command_encoder(&wgpu::CommandEncoderDescriptor { label: Some("Raymarching Render Encoder") });
        encoder.copy_buffer_to_texture(
            wgpu::ImageCopyBuffer { buffer: &self.staging, layout: wgpu::ImageDataLayout { offset: 0, bytes_per_row: Some(pitch as u32), rows_per_image: Some(self.height) } },
            wgpu::ImageCopyTexture { texture: &self.blit_texture, mip_level: 0, origin: wgpu::Origin3d::ZERO, aspect: wgpu::TextureAspect::All },
            wgpu::Extent3d { width: self.width, height: self.height, depth_or_array_layers: 1 });
        {
            let mut pass = encoder.begin_render_pass(&wgpu::RenderPassDescriptor {
                label: Some("Raymarching Render Pass"),
                color_attachments: &[Some(wgpu::RenderPassColorAttachment { view, resolve_target: None,
                    ops: wgpu::Operations { load: wgpu::LoadOp::Clear(Color::TRANSPARENT), store: wgpu::StoreOp::Store } })],   // mod.rs:124
                depth_stencil_attachment: None, occlusion_query_set: None, timestamp_writes: None,
            });
            pass.set_pipeline(&self.blit_pipeline);
            pass.set_bind_group(0, &self.blit_bind_group, &[]);
            pass.draw(0..3, 0..1);
        }
        encoder.finish()
    }
}

impl Drop for RayMarcher {
    fn drop(&mut self) {
        VRM_CTX.with(|c| *c.borrow_mut() = std::ptr::null_mut());
        unsafe {
            vrm_sys::vrm_image_destroy(self.target);
            vrm_sys::vrm_destroy(self.ctx);
        }
    }
}
