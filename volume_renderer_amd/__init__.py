"""volume_renderer_amd -- MI355X-native ray-marching path of bennetde/volume-renderer.

Python mirror of the reference's host interface for this path (same names and argument meaning as
the Rust types in /root/reference/src: RayMarcher, VoxelGrid, Screenshotter, Camera,
CameraSphereController, SphereScreenshotManager) over the C ABI in include/vrm.h.  All compute is in
libvrm.so (hand-written HIP for gfx950); there is no CPU path here.
"""
import ctypes
import os

import numpy as np

from . import _ffi
from ._ffi import (CameraUniform, GridUniform, TransferFunction, VrmError, OPT_CHUNK_STEPS, OPT_DIAGNOSTICS, OPT_KERNEL,
                   OPT_COMPACT_SCALAR, OPT_PACKED_ALPHA, OPT_PROFILE, OPT_RAYS_PER_LANE, OPT_SKIP_EMPTY)

__all__ = ["RayMarcher", "VoxelGrid", "Screenshotter", "Camera", "CameraSphereController", "SphereScreenshotManager",
           "CameraUniform", "TransferFunction", "GridUniform", "VrmError", "sphere_views"]


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


class Camera:
    """camera.rs:5-40 + transform.rs: position + look-at; fovy 45 deg, near 0.1, far 1000."""

    def __init__(self, aspect_ratio=1.0):
        self.position = np.array([-5.0, -5.0, -5.0], dtype=np.float32)  # camera.rs:17
        self.target = np.zeros(3, dtype=np.float32)
        self.aspect = float(aspect_ratio)
        self.fovy = 45.0

    def set_aspect_ratio(self, aspect_ratio):
        self.aspect = float(aspect_ratio)

    def look_to(self, pos):
        """Transform::look_to(pos, Vec3::Y) (transform.rs:59-63): aim at world point `pos`."""
        self.target = _f32(pos)

    def uniform(self, with_basis=False):
        """CameraUniform::update_view_proj (camera.rs:61-67)."""
        u = CameraUniform()
        basis = np.zeros(9, dtype=np.float32)
        pos, tgt = _f32(self.position), _f32(self.target)
        _ffi.check(_ffi.lib().vrm_camera_look_at(_fp(pos), _fp(tgt), ctypes.c_float(self.aspect), ctypes.byref(u), _fp(basis)))
        return (u, basis.reshape(3, 3)) if with_basis else u


def uniform_to_array(u):
    """CameraUniform -> 36 f32 (position, view_proj, inverse_view_proj)."""
    return np.frombuffer(bytes(u), dtype=np.float32).copy()


def uniform_from_array(a):
    a = _f32(a)
    assert a.size == 36
    return CameraUniform.from_buffer_copy(a.tobytes())


class CameraSphereController:
    """camera_sphere_controller.rs:9-94."""

    def __init__(self, y_divisions=8, x_divisions=8, origin=(0.0, 0.0, 0.0), radius=100.0):
        self.vertical_divisions = int(y_divisions)
        self.horizontal_divisions = int(x_divisions)
        self.current_index_x = 0
        self.current_index_y = 4  # camera_sphere_controller.rs:25
        self.radius = float(radius)
        self.origin = _f32(origin)

    def x_divisions(self):
        return self.horizontal_divisions

    def y_divisions(self):
        return self.vertical_divisions

    def get_position_on_sphere(self):
        out = np.zeros(3, dtype=np.float32)
        _ffi.check(_ffi.lib().vrm_sphere_position(self.current_index_x, self.current_index_y, self.horizontal_divisions,
                                                  self.vertical_divisions, ctypes.c_float(self.radius), _fp(self.origin), _fp(out)))
        return out

    def update_position(self, camera):
        camera.position = self.get_position_on_sphere()
        camera.look_to(self.origin)

    def get_position_as_string(self):
        return f"{self.current_index_x}-{self.current_index_y}"


def sphere_views(x_div, y_div):
    """View order of "Screenshot All" (sphere_screenshot_manager.rs:23-54): y outer from 1, x inner."""
    return [(ix, iy) for iy in range(1, y_div) for ix in range(x_div)]


class VoxelGrid:
    """voxel/grid.rs:9-22: the volume + its two uniforms, owned by a RayMarcher."""

    def __init__(self, ray_marcher):
        self._rm = ray_marcher
        self.dimensions = (3, 3, 3)
        self.attenuation = 1.0
        self.transfer_function_colors = self._default_tf()

    @staticmethod
    def _default_tf():
        tf = TransferFunction()
        tf.color_a[:] = (0.117, 0.188, 0.62, 1.0)  # grid.rs:285-287
        tf.color_b[:] = (0.7294, 0.7294, 0.7294, 1.0)
        tf.color_c[:] = (0.5725, 0.0, 0.039, 1.0)
        tf.use_transfer_function[:] = (0, 0, 0, 0)
        return tf

    def new(self, dimensions):
        """VoxelGrid::new(dimensions, ..): fresh grid of [0,0,0,255]; attenuation and TF reset (grid.rs:197-198)."""
        x, y, z = (int(d) for d in dimensions)
        _ffi.check(_ffi.lib().vrm_grid_resize(self._rm._h, x, y, z))
        self.dimensions = (x, y, z)
        self.attenuation = 1.0
        self.transfer_function_colors = self._default_tf()

    def new_slab(self, dimensions, z_begin, z_end):
        """One z slab of a `dimensions` volume for sort-last rendering (vrm_grid_resize_slab): owns the slices
        [z_begin, z_end), stores one ghost slice either side.  Afterwards `dimensions` is the STORED shape the upload
        calls expect; returns (z_store_begin, z_store_end)."""
        x, y, z = (int(d) for d in dimensions)
        _ffi.check(_ffi.lib().vrm_grid_resize_slab(self._rm._h, x, y, z, int(z_begin), int(z_end)))
        s0, s1 = self.stored_slices()
        self.dimensions = (x, y, s1 - s0)
        self.attenuation = 1.0
        self.transfer_function_colors = self._default_tf()
        return s0, s1

    def stored_slices(self):
        s0, s1 = ctypes.c_uint32(), ctypes.c_uint32()
        _ffi.check(_ffi.lib().vrm_grid_get_slab(self._rm._h, ctypes.byref(s0), ctypes.byref(s1), None, None))
        return s0.value, s1.value

    def upload_rgba8(self, voxels_zyxc):
        """set_color over the grid + update_buffer (grid.rs:202-238). voxels: uint8 (Z,Y,X,4)."""
        v = np.ascontiguousarray(voxels_zyxc, dtype=np.uint8)
        assert v.ndim == 4 and v.shape[3] == 4
        if (v.shape[2], v.shape[1], v.shape[0]) != tuple(self.dimensions):
            self.new((v.shape[2], v.shape[1], v.shape[0]))
        _ffi.check(_ffi.lib().vrm_grid_upload_rgba8(self._rm._h, v.ctypes.data, v.nbytes))

    def upload_rgba8_device(self, device_ptr, nbytes):
        _ffi.check(_ffi.lib().vrm_grid_upload_rgba8_device(self._rm._h, device_ptr, nbytes))

    def upload_f32(self, data):
        """loaders/netcdf.rs:45-61. data: f32 (4,Z,Y,X) or (Z,Y,X) for a scalar field."""
        d = _f32(data)
        ch = 4 if d.ndim == 4 else 1
        shape = d.shape[1:] if d.ndim == 4 else d.shape
        if (shape[2], shape[1], shape[0]) != tuple(self.dimensions):
            self.new((shape[2], shape[1], shape[0]))
        _ffi.check(_ffi.lib().vrm_grid_upload_f32(self._rm._h, d.ctypes.data, ch))

    def upload_scalar(self, raw, dimensions, bytes_per_sample=1):
        """loaders/dat.rs:58-76 on an in-memory RAW block."""
        r = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1)
        if tuple(dimensions) != tuple(self.dimensions):
            self.new(dimensions)
        _ffi.check(_ffi.lib().vrm_grid_upload_scalar(self._rm._h, r.ctypes.data, r.nbytes, bytes_per_sample))

    def resident_bytes(self):
        """device bytes the grid holds (vrm_grid_resident_bytes): 9 B/voxel, or about 1 B/voxel for a compact scalar volume"""
        n = ctypes.c_size_t()
        _ffi.check(_ffi.lib().vrm_grid_resident_bytes(self._rm._h, ctypes.byref(n)))
        return int(n.value)

    def download_rgba8(self):
        x, y, _ = (int(d) for d in self.uniform().dimensions[:3])
        s0, s1 = self.stored_slices()
        z = s1 - s0                                   # the stored slices (all of them unless slab-partitioned)
        self.dimensions = (x, y, z)
        out = np.zeros((z, y, x, 4), dtype=np.uint8)
        _ffi.check(_ffi.lib().vrm_grid_download_rgba8(self._rm._h, out.ctypes.data, out.nbytes))
        return out

    def update_voxel_grid_buffer(self):
        """grid.rs:240-242: push self.attenuation."""
        _ffi.check(_ffi.lib().vrm_set_attenuation(self._rm._h, ctypes.c_float(self.attenuation)))

    def update_transfer_function_buffer(self):
        """grid.rs:244-246: push self.transfer_function_colors."""
        _ffi.check(_ffi.lib().vrm_set_transfer_function(self._rm._h, ctypes.byref(self.transfer_function_colors)))

    def set_transfer_function_active(self, val):
        self.transfer_function_colors.use_transfer_function[0] = 1 if val else 0

    def uniform(self):
        u = GridUniform()
        _ffi.check(_ffi.lib().vrm_get_grid_uniform(self._rm._h, ctypes.byref(u)))
        return u


class Screenshotter:
    """screenshot.rs:6-97 plus the render target it copies from."""

    def __init__(self, ray_marcher, width, height):
        self._rm = ray_marcher
        self.width, self.height = int(width), int(height)
        h = ctypes.c_void_p()
        _ffi.check(_ffi.lib().vrm_image_create(ray_marcher._h, self.width, self.height, ctypes.byref(h)))
        self._h = h

    def close(self):
        if self._h:
            _ffi.lib().vrm_image_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def screenshot(self):
        """screenshot.rs:30-54: enqueue the device->host copy."""
        _ffi.check(_ffi.lib().vrm_image_readback_async(self._h))

    def read(self):
        """map + copy out + unmap: (H, W, 4) uint8 without row padding."""
        p = ctypes.POINTER(ctypes.c_uint8)()
        pitch = ctypes.c_size_t()
        _ffi.check(_ffi.lib().vrm_image_map(self._h, ctypes.byref(p), ctypes.byref(pitch)))
        try:
            buf = np.ctypeslib.as_array(p, shape=(self.height, pitch.value))
            out = buf[:, : self.width * 4].reshape(self.height, self.width, 4).copy()
        finally:
            _ffi.check(_ffi.lib().vrm_image_unmap(self._h))
        return out

    def wait(self):
        """map + unmap without copying out: returns once the last readback has landed in host memory."""
        p = ctypes.POINTER(ctypes.c_uint8)()
        pitch = ctypes.c_size_t()
        _ffi.check(_ffi.lib().vrm_image_map(self._h, ctypes.byref(p), ctypes.byref(pitch)))
        _ffi.check(_ffi.lib().vrm_image_unmap(self._h))

    def save_screenshot_to_disk(self, file_name):
        """screenshot.rs:57-92."""
        _ffi.check(_ffi.lib().vrm_image_save_png(self._h, os.fsencode(file_name)))

    def diagnostics(self):
        rgba = np.zeros((self.height, self.width, 4), dtype=np.float32)
        it = np.zeros((self.height, self.width), dtype=np.uint32)
        _ffi.check(_ffi.lib().vrm_image_read_diagnostics(self._h, _fp(rgba), it.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
        return rgba, it

    def executed_samples(self):
        """(H, W) u32: samples actually filtered per ray in the last draw with OPT_DIAGNOSTICS on (N_exec)."""
        n = np.zeros((self.height, self.width), dtype=np.uint32)
        _ffi.check(_ffi.lib().vrm_image_read_sample_counts(self._h, n.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))))
        return n

    def ray_diagnostics(self):
        """(H, W, 8) f32: rd.xyz, t_min, t_max, n_inside, hit, 0."""
        g = np.zeros((self.height, self.width, 8), dtype=np.float32)
        _ffi.check(_ffi.lib().vrm_image_read_ray_diagnostics(self._h, _fp(g)))
        return g


class RayMarcher:
    """ray_marcher/mod.rs:18-142: owns the VoxelGrid, draws views into targets."""

    def __init__(self, device_id=0, width=1024, height=1024):
        cfg = _ffi.Config(int(device_id), int(width), int(height))
        h = ctypes.c_void_p()
        _ffi.check(_ffi.lib().vrm_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        self.width, self.height = int(width), int(height)
        self.voxel_grid = VoxelGrid(self)

    def close(self):
        if self._h:
            _ffi.lib().vrm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def draw(self, camera_uniform, target):
        """RayMarcher::draw: enqueue only."""
        _ffi.check(_ffi.lib().vrm_draw(self._h, ctypes.byref(camera_uniform), target._h))

    def draw_batch(self, camera_uniforms, targets):
        n = len(camera_uniforms)
        assert n == len(targets) and n > 0
        cams = (CameraUniform * n)(*camera_uniforms)
        hs = (ctypes.c_void_p * n)(*[t._h for t in targets])
        _ffi.check(_ffi.lib().vrm_draw_batch(self._h, cams, n, hs))

    def draw_batch_static(self, camera_uniforms, targets):
        """vrm_draw_batch_static: the same draw with the kernel's descriptor scratch in a static device array -- no
        allocation of any kind inside the call; at most 256 views."""
        n = len(camera_uniforms)
        assert n == len(targets) and n > 0
        cams = (CameraUniform * n)(*camera_uniforms)
        hs = (ctypes.c_void_p * n)(*[t._h for t in targets])
        _ffi.check(_ffi.lib().vrm_draw_batch_static(self._h, cams, n, hs))

    # -- the volume broadcast of a view-partitioned export inside the library (RCCL bound at first use; include/vrm.h) --
    @staticmethod
    def comm_unique_id():
        """128 bytes made on rank 0; hand them to the other ranks (file, socket, torch.distributed object broadcast ...)."""
        buf = (ctypes.c_uint8 * 128)()
        _ffi.check(_ffi.lib().vrm_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, comm_id, rank, n_ranks):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(comm_id)
        _ffi.check(_ffi.lib().vrm_comm_init(self._h, buf, int(rank), int(n_ranks)))

    def grid_broadcast(self, root=0):
        """Every rank holds a grid of the same dimensions; the root's voxels arrive in all of them."""
        _ffi.check(_ffi.lib().vrm_grid_broadcast(self._h, int(root)))

    def comm_destroy(self):
        _ffi.check(_ffi.lib().vrm_comm_destroy(self._h))

    def draw_partial(self, camera_uniforms, width, height, device_ptr):
        """n views as premultiplied f32 RGBA into caller-owned device memory (n*H*W*4 floats); enqueue only.  For a slab
        grid (VoxelGrid.new_slab) this is the slab's partial image of a sort-last render."""
        n = len(camera_uniforms)
        cams = (CameraUniform * n)(*camera_uniforms)
        _ffi.check(_ffi.lib().vrm_draw_partial(self._h, cams, n, int(width), int(height), int(device_ptr)))

    def composite(self, camera_uniform, partials_ptr, n_slabs, width, height, row0, rows, out_rgba8_ptr=None, out_f32_ptr=None):
        """Over-composite n_slabs partials ([slab][row][x][4] f32, device) of rows [row0, row0+rows) of one view; enqueue only."""
        _ffi.check(_ffi.lib().vrm_composite(self._h, ctypes.byref(camera_uniform), int(partials_ptr), int(n_slabs), int(width),
                                            int(height), int(row0), int(rows), out_rgba8_ptr, out_f32_ptr))

    def count_reference_samples(self, camera_uniforms, width, height):
        """loop iterations the reference would execute for these views (unit of Gsamples/s)."""
        n = len(camera_uniforms)
        cams = (CameraUniform * n)(*camera_uniforms)
        total = ctypes.c_uint64()
        _ffi.check(_ffi.lib().vrm_count_reference_samples(self._h, cams, n, int(width), int(height), ctypes.byref(total)))
        return total.value

    def diag_counters(self, reset=True):
        """Counters of the diagnostics draws since the last reset (vrm_diag_read_counters) as a dict."""
        out = (ctypes.c_uint64 * 8)()
        _ffi.check(_ffi.lib().vrm_diag_read_counters(self._h, out, 1 if reset else 0))
        names = ["executed", "executed_zero_alpha", "chunks", "chunks_staged", "chunks_empty", "probes_failed", "chunks_gathered",
                 "box_retries"]
        return {k: int(out[i]) for i, k in enumerate(names)}

    def synchronize(self):
        _ffi.check(_ffi.lib().vrm_synchronize(self._h))

    def set_option(self, option, value):
        _ffi.check(_ffi.lib().vrm_set_option(self._h, int(option), int(value)))

    def get_option(self, option):
        v = ctypes.c_int64()
        _ffi.check(_ffi.lib().vrm_get_option(self._h, int(option), ctypes.byref(v)))
        return v.value

    def profile_read(self, reset=True):
        ms = ctypes.c_double()
        n = ctypes.c_uint64()
        _ffi.check(_ffi.lib().vrm_profile_read(self._h, ctypes.byref(ms), ctypes.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def load_dat(self, path):
        """loaders::dat::open_voxel_grid (dat.rs:10-82)."""
        _ffi.check(_ffi.lib().vrm_load_dat(self._h, os.fsencode(path)))
        u = self.voxel_grid.uniform()
        self.voxel_grid.dimensions = tuple(int(d) for d in u.dimensions[:3])
        self.voxel_grid.attenuation = 1.0
        self.voxel_grid.transfer_function_colors = VoxelGrid._default_tf()

    def load_netcdf(self, path):
        """loaders::netcdf::open_voxel_grid (netcdf.rs:34-69); classic NetCDF containers (see include/vrm.h)."""
        _ffi.check(_ffi.lib().vrm_load_netcdf(self._h, os.fsencode(path)))
        u = self.voxel_grid.uniform()
        self.voxel_grid.dimensions = tuple(int(d) for d in u.dimensions[:3])
        self.voxel_grid.attenuation = 1.0
        self.voxel_grid.transfer_function_colors = VoxelGrid._default_tf()

    def write_netcdf(self, path, format="classic"):
        """loaders::netcdf::write_voxel_grid (netcdf.rs:8-31); `format` "classic" (CDF-1/2/5) or "netcdf4" (the HDF5 container
        the reference's `netcdf::create` makes)."""
        if format not in ("classic", "netcdf4"):
            raise ValueError(f"format is 'classic' or 'netcdf4', got {format!r}")
        fn = _ffi.lib().vrm_write_netcdf if format == "classic" else _ffi.lib().vrm_write_netcdf4
        _ffi.check(fn(self._h, os.fsencode(path)))

    def compare_to_netcdf_rmse(self, color_czyx, density_only):
        """compare::netcdf::compare_to_netcdf_rmse (compare/netcdf.rs:8-75) against an in-memory `color(c,z,y,x)` f32 array."""
        d = _f32(color_czyx)
        X, Y, Z = self.voxel_grid.dimensions
        if d.size != 4 * X * Y * Z:
            raise ValueError(f"expected color(4,{Z},{Y},{X}), got {d.shape}")
        out = ctypes.c_float()
        _ffi.check(_ffi.lib().vrm_compare_volume_f32(self._h, d.ctypes.data, 1 if density_only else 0, ctypes.byref(out)))
        return out.value

    def ssd_u8(self, a, b):
        """Exact sum of squared differences of two uint8 arrays (GPU reduction); mse = ssd / a.size (similarity.py:61)."""
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        if a.shape != b.shape:
            raise ValueError(f"shapes differ: {a.shape} vs {b.shape}")
        out = ctypes.c_uint64()
        _ffi.check(_ffi.lib().vrm_ssd_u8(self._h, a.ctypes.data, b.ctypes.data, a.size, ctypes.byref(out)))
        return out.value

    def ssim_u8(self, a, b, data_range):
        """skimage.metrics.structural_similarity(a, b, data_range=data_range, channel_axis=-1) for two uint8 images
        of shape (h, w) or (h, w, c), c <= 4 (similarity.py:65; GPU reduction, scikit-image defaults restated)."""
        a = np.ascontiguousarray(a, dtype=np.uint8)
        b = np.ascontiguousarray(b, dtype=np.uint8)
        if a.shape != b.shape or a.ndim not in (2, 3):
            raise ValueError(f"need two (h, w[, c]) images of one shape, got {a.shape} and {b.shape}")
        h, w = a.shape[:2]
        c = a.shape[2] if a.ndim == 3 else 1
        out = ctypes.c_double()
        _ffi.check(_ffi.lib().vrm_ssim_u8(self._h, a.ctypes.data, b.ctypes.data, w, h, c, float(data_range), ctypes.byref(out)))
        return out.value

    def image_ssd(self, img_a, img_b):
        """SSD of two rendered images on the device -> (ssd, number of channel values)."""
        ssd, cnt = ctypes.c_uint64(), ctypes.c_uint64()
        _ffi.check(_ffi.lib().vrm_image_ssd(img_a._h, img_b._h, ctypes.byref(ssd), ctypes.byref(cnt)))
        return ssd.value, cnt.value

    def export_sphere(self, x_div, y_div, radius, out_dir, origin=(0.0, 0.0, 0.0), part=0, n_parts=1):
        """"Screenshot All" (sphere_screenshot_manager.rs:23-73, state.rs:391-397)."""
        org = _f32(origin)
        os.makedirs(out_dir, exist_ok=True)
        _ffi.check(_ffi.lib().vrm_export_sphere(self._h, x_div, y_div, ctypes.c_float(radius), _fp(org), os.fsencode(out_dir),
                                                part, n_parts))


class SphereScreenshotManager:
    """sphere_screenshot_manager.rs:9-75 as an iterator over (ix, iy, CameraUniform)."""

    def __init__(self, csp):
        self.csp = csp

    def views(self, aspect=1.0):
        cam = Camera(aspect)
        for ix, iy in sphere_views(self.csp.x_divisions(), self.csp.y_divisions()):
            self.csp.current_index_x, self.csp.current_index_y = ix, iy
            self.csp.update_position(cam)
            yield ix, iy, cam.uniform()
