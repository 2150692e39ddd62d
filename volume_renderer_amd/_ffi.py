"""ctypes binding of libvrm.so (include/vrm.h).  No fallback: a missing library or GPU raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VRM_LIB: another build of the same library (kernel A/B runs of tools/ab_builds.sh); the default is the in-tree build
LIB_PATH = os.environ.get("VRM_LIB") or os.path.join(_HERE, "libvrm.so")

VRM_OK = 0
VRM_ERR_INVALID = -1
VRM_ERR_NO_DEVICE = -2
VRM_ERR_HIP = -3
VRM_ERR_IO = -4
VRM_ERR_STATE = -5

OPT_KERNEL = 0
OPT_SKIP_EMPTY = 1
OPT_PROFILE = 2
OPT_DIAGNOSTICS = 3
OPT_CHUNK_STEPS = 4
OPT_PACKED_ALPHA = 5
OPT_COMPACT_SCALAR = 6
OPT_RAYS_PER_LANE = 7


class VrmError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"vrm status {status}: {message}")
        self.status = status


class CameraUniform(ctypes.Structure):
    """CameraUniform, reference src/camera.rs:42-50 (144 bytes)."""
    _fields_ = [("position", ctypes.c_float * 4), ("view_proj", ctypes.c_float * 16),
                ("inverse_view_proj", ctypes.c_float * 16)]


class TransferFunction(ctypes.Structure):
    """RaymarchTransferFunctionColors, reference src/voxel/grid.rs:34-41 (64 bytes)."""
    _fields_ = [("color_a", ctypes.c_float * 4), ("color_b", ctypes.c_float * 4), ("color_c", ctypes.c_float * 4),
                ("use_transfer_function", ctypes.c_uint32 * 4)]


class GridUniform(ctypes.Structure):
    """VoxelGridUniform, reference src/voxel/grid.rs:24-32 (64 bytes)."""
    _fields_ = [("dimensions", ctypes.c_uint32 * 4), ("box_min", ctypes.c_float * 4), ("box_size", ctypes.c_float * 4),
                ("buffer", ctypes.c_float * 4)]


class Config(ctypes.Structure):
    _fields_ = [("device_id", ctypes.c_int32), ("width", ctypes.c_uint32), ("height", ctypes.c_uint32)]


# every symbol include/vrm.h declares: name -> (restype, argtypes)
_vp = ctypes.c_void_p
_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
SYMBOLS = {
    "vrm_create": (ctypes.c_int, [ctypes.POINTER(Config), ctypes.POINTER(_vp)]),
    "vrm_destroy": (None, [_vp]),
    "vrm_draw": (ctypes.c_int, [_vp, ctypes.POINTER(CameraUniform), _vp]),
    "vrm_draw_batch": (ctypes.c_int, [_vp, ctypes.POINTER(CameraUniform), ctypes.c_uint32, ctypes.POINTER(_vp)]),
    "vrm_draw_batch_static": (ctypes.c_int, [_vp, ctypes.POINTER(CameraUniform), ctypes.c_uint32, ctypes.POINTER(_vp)]),
    "vrm_synchronize": (ctypes.c_int, [_vp]),
    "vrm_grid_resize": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]),
    "vrm_grid_upload_rgba8": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "vrm_grid_upload_rgba8_device": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "vrm_grid_upload_f32": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32]),
    "vrm_grid_upload_f32_device": (ctypes.c_int, [_vp, _vp, ctypes.c_uint32]),
    "vrm_grid_upload_scalar": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t, ctypes.c_uint32]),
    "vrm_grid_download_rgba8": (ctypes.c_int, [_vp, _vp, ctypes.c_size_t]),
    "vrm_set_attenuation": (ctypes.c_int, [_vp, ctypes.c_float]),
    "vrm_set_transfer_function": (ctypes.c_int, [_vp, ctypes.POINTER(TransferFunction)]),
    "vrm_get_grid_uniform": (ctypes.c_int, [_vp, ctypes.POINTER(GridUniform)]),
    "vrm_image_create": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(_vp)]),
    "vrm_image_destroy": (None, [_vp]),
    "vrm_image_readback_async": (ctypes.c_int, [_vp]),
    "vrm_image_map": (ctypes.c_int, [_vp, ctypes.POINTER(_u8p), ctypes.POINTER(ctypes.c_size_t)]),
    "vrm_image_unmap": (ctypes.c_int, [_vp]),
    "vrm_image_save_png": (ctypes.c_int, [_vp, ctypes.c_char_p]),
    "vrm_camera_look_at": (ctypes.c_int, [_f32p, _f32p, ctypes.c_float, ctypes.POINTER(CameraUniform), _f32p]),
    "vrm_sphere_position": (ctypes.c_int, [ctypes.c_uint32] * 4 + [ctypes.c_float, _f32p, _f32p]),
    "vrm_export_sphere": (ctypes.c_int, [_vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_float, _f32p, ctypes.c_char_p,
                                         ctypes.c_uint32, ctypes.c_uint32]),
    "vrm_load_dat": (ctypes.c_int, [_vp, ctypes.c_char_p]),
    "vrm_set_option": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int64]),
    "vrm_get_option": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]),
    "vrm_profile_read": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]),
    "vrm_image_read_diagnostics": (ctypes.c_int, [_vp, _f32p, _u32p]),
    "vrm_image_read_sample_counts": (ctypes.c_int, [_vp, _u32p]),
    "vrm_image_read_ray_diagnostics": (ctypes.c_int, [_vp, _f32p]),
    "vrm_diag_read_counters": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]),
    "vrm_count_reference_samples": (ctypes.c_int, [_vp, ctypes.POINTER(CameraUniform), ctypes.c_uint32, ctypes.c_uint32,
                                                   ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)]),
    "vrm_comm_unique_id": (ctypes.c_int, [_vp]),
    "vrm_comm_init": (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int]),
    "vrm_grid_broadcast": (ctypes.c_int, [_vp, ctypes.c_int]),
    "vrm_comm_destroy": (ctypes.c_int, [_vp]),
    "vrm_load_netcdf": (ctypes.c_int, [_vp, ctypes.c_char_p]),
    "vrm_write_netcdf": (ctypes.c_int, [_vp, ctypes.c_char_p]),
    "vrm_write_netcdf4": (ctypes.c_int, [_vp, ctypes.c_char_p]),
    "vrm_netcdf_read_color": (ctypes.c_int, [ctypes.c_char_p, _u32p, _u32p, _u32p, _vp, ctypes.c_size_t]),
    "vrm_netcdf_write_color": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _vp]),
    "vrm_netcdf_write_color_nc4": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _vp]),
    "vrm_grid_resize_slab": (ctypes.c_int, [_vp] + [ctypes.c_uint32] * 5),
    "vrm_grid_get_slab": (ctypes.c_int, [_vp, _u32p, _u32p, _u32p, _u32p]),
    "vrm_grid_resident_bytes": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_size_t)]),
    "vrm_draw_partial": (ctypes.c_int, [_vp, ctypes.POINTER(CameraUniform), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, _vp]),
    "vrm_composite": (ctypes.c_int, [_vp, ctypes.POINTER(CameraUniform), _vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                     ctypes.c_uint32, ctypes.c_uint32, _vp, _vp]),
    "vrm_compare_volume_f32": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _f32p]),
    "vrm_compare_volume_f32_device": (ctypes.c_int, [_vp, _vp, ctypes.c_int, _f32p]),
    "vrm_image_ssd": (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]),
    "vrm_ssd_u8": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]),
    "vrm_ssim_u8": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double,
                                   ctypes.POINTER(ctypes.c_double)]),
    "vrm_png_read": (ctypes.c_int, [ctypes.c_char_p, _u32p, _u32p, _u32p, _vp, ctypes.c_size_t]),
    "vrm_last_error": (ctypes.c_char_p, []),
    "vrm_version": (ctypes.c_char_p, []),
}

_lib = None


def lib():
    """Load libvrm.so (built in-tree by `make -C volume_renderer_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `make -C volume_renderer_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != VRM_OK:
        raise VrmError(status, lib().vrm_last_error().decode("utf-8", "replace"))
