"""ctypes loader for the CPU oracle (oracle/vr_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (volume_renderer_amd).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libvr_oracle.so")
_lib = None

# RaymarchTransferFunctionColors::new (reference src/voxel/grid.rs:282-290)
DEFAULT_TF_COLORS = (
    (0.117, 0.188, 0.62, 1.0),
    (0.7294, 0.7294, 0.7294, 1.0),
    (0.5725, 0.0, 0.039, 1.0),
)


def build(force=False):
    """Compile libvr_oracle.so with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "vr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libvr_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        f32p = ctypes.POINTER(ctypes.c_float)
        u32p = ctypes.POINTER(ctypes.c_uint32)
        u8p = ctypes.POINTER(ctypes.c_uint8)
        L.orc_camera_uniform.argtypes = [f32p, f32p, ctypes.c_float, f32p, f32p]
        L.orc_camera_uniform.restype = None
        L.orc_sphere_position.argtypes = [ctypes.c_uint32] * 4 + [ctypes.c_float, f32p, f32p]
        L.orc_sphere_position.restype = None
        L.orc_grid_uniform.argtypes = [u32p, ctypes.c_float, u32p, f32p, f32p, f32p]
        L.orc_grid_uniform.restype = None
        L.orc_quantize_dat.argtypes = [u8p, ctypes.c_size_t, ctypes.c_int, u8p]
        L.orc_quantize_dat.restype = None
        L.orc_quantize_netcdf.argtypes = [f32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, u8p]
        L.orc_quantize_netcdf.restype = None
        L.orc_render.argtypes = [u8p, u32p, f32p, u32p, ctypes.c_float, ctypes.c_uint32, ctypes.c_uint32,
                                 f32p, u8p, u32p, ctypes.c_int]
        L.orc_render.restype = ctypes.c_uint64
        L.orc_scene.argtypes = [u8p, u32p, f32p, f32p]
        L.orc_scene.restype = None
        L.orc_ray.argtypes = [f32p, u32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                              f32p, f32p, f32p]
        L.orc_ray.restype = ctypes.c_int
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_geometry.argtypes = [f32p, u32p, ctypes.c_uint32, ctypes.c_uint32, f32p]
        L.orc_geometry.restype = None
        L.orc_render_slab.argtypes = [u8p, u32p, f32p, u32p, ctypes.c_float, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                      ctypes.c_uint32, f32p, ctypes.c_int]
        L.orc_render_slab.restype = ctypes.c_uint64
        L.orc_render_slab_stored.argtypes = [u8p, ctypes.c_int, u32p, ctypes.c_uint32, ctypes.c_uint32, f32p, u32p, ctypes.c_float,
                                             ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, u32p, f32p, u32p,
                                             ctypes.c_int]
        L.orc_render_slab_stored.restype = ctypes.c_uint64
        L.orc_compare_volume.argtypes = [u8p, f32p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, f32p, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_double)]
        L.orc_compare_volume.restype = ctypes.c_float
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(ctypes.POINTER(ty))


def camera_uniform(position, target=(0.0, 0.0, 0.0), aspect=1.0, with_basis=False):
    """CameraUniform (36 f32: position[4], view_proj[16], inverse_view_proj[16]) for a camera at
    `position` aimed at `target` with up=+Y (reference camera.rs:61-67, transform.rs:59-63)."""
    pos = np.asarray(position, dtype=np.float32)
    tgt = np.asarray(target, dtype=np.float32)
    out = np.zeros(36, dtype=np.float32)
    basis = np.zeros(9, dtype=np.float32)
    lib().orc_camera_uniform(_p(pos, ctypes.c_float), _p(tgt, ctypes.c_float), ctypes.c_float(aspect),
                             _p(out, ctypes.c_float), _p(basis, ctypes.c_float))
    return (out, basis.reshape(3, 3)) if with_basis else out


def sphere_position(ix, iy, xdiv, ydiv, radius, origin=(0.0, 0.0, 0.0)):
    org = np.asarray(origin, dtype=np.float32)
    out = np.zeros(3, dtype=np.float32)
    lib().orc_sphere_position(ix, iy, xdiv, ydiv, ctypes.c_float(radius), _p(org, ctypes.c_float), _p(out, ctypes.c_float))
    return out


def sphere_views(xdiv, ydiv):
    """View order of the export loop (sphere_screenshot_manager.rs:23-54): y outer 1..ydiv-1, x inner."""
    return [(ix, iy) for iy in range(1, ydiv) for ix in range(xdiv)]


def grid_uniform(dims, attenuation=1.0):
    d = np.asarray(dims, dtype=np.uint32)
    od = np.zeros(4, dtype=np.uint32)
    bmin = np.zeros(4, dtype=np.float32)
    bsz = np.zeros(4, dtype=np.float32)
    buf = np.zeros(4, dtype=np.float32)
    lib().orc_grid_uniform(_p(d, ctypes.c_uint32), ctypes.c_float(attenuation), _p(od, ctypes.c_uint32),
                           _p(bmin, ctypes.c_float), _p(bsz, ctypes.c_float), _p(buf, ctypes.c_float))
    return od, bmin, bsz, buf


def tf_words(use_tf=False, colors=DEFAULT_TF_COLORS):
    """RaymarchTransferFunctionColors as 16 32-bit words (grid.rs:34-41)."""
    w = np.zeros(16, dtype=np.uint32)
    w[:12] = np.asarray(colors, dtype=np.float32).reshape(12).view(np.uint32)
    w[12] = 1 if use_tf else 0
    return w


def quantize_dat(raw, bytes_per_sample=1):
    raw = np.ascontiguousarray(raw, dtype=np.uint8).reshape(-1)
    n = raw.size // bytes_per_sample
    out = np.zeros(n * 4, dtype=np.uint8)
    lib().orc_quantize_dat(_p(raw, ctypes.c_uint8), n, bytes_per_sample, _p(out, ctypes.c_uint8))
    return out.reshape(n, 4)


def quantize_netcdf(data_czyx):
    """f32 color(c=4,z,y,x) -> RGBA8 voxels shaped (Z,Y,X,4) (x fastest)."""
    d = np.ascontiguousarray(data_czyx, dtype=np.float32)
    _, Z, Y, X = d.shape
    out = np.zeros((Z, Y, X, 4), dtype=np.uint8)
    lib().orc_quantize_netcdf(_p(d, ctypes.c_float), X, Y, Z, _p(out, ctypes.c_uint8))
    return out


def render(voxels_zyxc, cam36, width, height, attenuation=1.0, tf=None, want_f32=True, want_steps=False, threads=None):
    """Render one view. voxels_zyxc: uint8 array shaped (Z,Y,X,4). Returns dict(rgba8, f32, steps, samples)."""
    v = np.ascontiguousarray(voxels_zyxc, dtype=np.uint8)
    Z, Y, X, C = v.shape
    assert C == 4
    dims = np.array([X, Y, Z], dtype=np.uint32)
    cam = np.ascontiguousarray(cam36, dtype=np.float32)
    tfw = tf_words() if tf is None else np.ascontiguousarray(tf, dtype=np.uint32)
    rgba8 = np.zeros((height, width, 4), dtype=np.uint8)
    f32 = np.zeros((height, width, 4), dtype=np.float32) if want_f32 else None
    steps = np.zeros((height, width), dtype=np.uint32) if want_steps else None
    if threads is None:
        threads = min(lib().orc_max_threads(), os.cpu_count() or 1)
    total = lib().orc_render(_p(v, ctypes.c_uint8), _p(dims, ctypes.c_uint32), _p(cam, ctypes.c_float),
                             _p(tfw, ctypes.c_uint32), ctypes.c_float(attenuation), width, height,
                             _p(f32, ctypes.c_float) if want_f32 else None, _p(rgba8, ctypes.c_uint8),
                             _p(steps, ctypes.c_uint32) if want_steps else None, int(threads))
    return {"rgba8": rgba8, "f32": f32, "steps": steps, "samples": int(total)}


def scene(voxels_zyxc, p):
    v = np.ascontiguousarray(voxels_zyxc, dtype=np.uint8)
    Z, Y, X, _ = v.shape
    dims = np.array([X, Y, Z], dtype=np.uint32)
    pp = np.asarray(p, dtype=np.float32)
    out = np.zeros(4, dtype=np.float32)
    lib().orc_scene(_p(v, ctypes.c_uint8), _p(dims, ctypes.c_uint32), _p(pp, ctypes.c_float), _p(out, ctypes.c_float))
    return out


def ray(cam36, dims_xyz, px, py, width, height):
    cam = np.ascontiguousarray(cam36, dtype=np.float32)
    dims = np.asarray(dims_xyz, dtype=np.uint32)
    rd = np.zeros(3, dtype=np.float32)
    tmin = ctypes.c_float()
    tmax = ctypes.c_float()
    hit = lib().orc_ray(_p(cam, ctypes.c_float), _p(dims, ctypes.c_uint32), px, py, width, height, _p(rd, ctypes.c_float),
                        ctypes.byref(tmin), ctypes.byref(tmax))
    return bool(hit), rd, tmin.value, tmax.value


def geometry(cam36, dims_xyz, width, height):
    """(H, W, 8) f32 per ray: rd.xyz, t_min, t_max, n_inside (no early termination), hit, iterations."""
    cam = np.ascontiguousarray(cam36, dtype=np.float32)
    dims = np.asarray(dims_xyz, dtype=np.uint32)
    out = np.zeros((height, width, 8), dtype=np.float32)
    lib().orc_geometry(_p(cam, ctypes.c_float), _p(dims, ctypes.c_uint32), width, height, _p(out, ctypes.c_float))
    return out


def compare_volume(voxels_zyxc, data_czyx, tf_colors=DEFAULT_TF_COLORS, density_only=False):
    """compare_to_netcdf_rmse (compare/netcdf.rs:8-75) -> (rmse as the reference's f32 running sum gives it,
    the same with the sum carried in f64)."""
    v = np.ascontiguousarray(voxels_zyxc, dtype=np.uint8)
    Z, Y, X, _ = v.shape
    d = np.ascontiguousarray(data_czyx, dtype=np.float32)
    assert d.shape == (4, Z, Y, X), d.shape
    tf9 = np.ascontiguousarray(np.asarray(tf_colors, dtype=np.float32)[:, :3]).reshape(9)
    f64 = ctypes.c_double()
    r = lib().orc_compare_volume(_p(v, ctypes.c_uint8), _p(d, ctypes.c_float), X, Y, Z, _p(tf9, ctypes.c_float),
                                 1 if density_only else 0, ctypes.byref(f64))
    return float(r), float(f64.value)


def render_slab(voxels_zyxc, cam36, width, height, z_begin, z_end, attenuation=1.0, tf=None, threads=None):
    """Partial premultiplied f32 RGBA image (H, W, 4) of the z slab [z_begin, z_end) of a sort-last render."""
    v = np.ascontiguousarray(voxels_zyxc, dtype=np.uint8)
    Z, Y, X, _ = v.shape
    dims = np.array([X, Y, Z], dtype=np.uint32)
    cam = np.ascontiguousarray(cam36, dtype=np.float32)
    tfw = tf_words(False) if tf is None else np.ascontiguousarray(tf, dtype=np.uint32)
    out = np.zeros((height, width, 4), dtype=np.float32)
    nt = threads if threads is not None else max(1, min(16, os.cpu_count() or 1))
    lib().orc_render_slab(_p(v, ctypes.c_uint8), _p(dims, ctypes.c_uint32), _p(cam, ctypes.c_float), _p(tfw, ctypes.c_uint32),
                          ctypes.c_float(attenuation), width, height, int(z_begin), int(z_end), _p(out, ctypes.c_float), nt)
    return out


def render_slab_stored(stored, dims_xyz, z_store0, cam36, width, height, z_begin, z_end, crop=None, attenuation=1.0, tf=None,
                       threads=None):
    """render_slab from the stored slices alone: `stored` is uint8 (S, Y, X) (scalar density, [b,b,b,b]) or (S, Y, X, 4)
    holding the slices [z_store0, z_store0 + S) of a dims_xyz volume; crop = (x0, y0, w, h) pixels of the width x height
    image (default: all).  Returns (f32 partial (h, w, 4), loop iterations per ray (h, w))."""
    v = np.ascontiguousarray(stored, dtype=np.uint8)
    scalar = v.ndim == 3
    S, Y, X = v.shape[:3]
    dims = np.asarray(dims_xyz, dtype=np.uint32)
    assert (X, Y) == (int(dims[0]), int(dims[1])) and z_store0 + S <= int(dims[2])
    crop = (0, 0, width, height) if crop is None else tuple(int(c) for c in crop)
    cr = np.asarray(crop, dtype=np.uint32)
    cam = np.ascontiguousarray(cam36, dtype=np.float32)
    tfw = tf_words(False) if tf is None else np.ascontiguousarray(tf, dtype=np.uint32)
    out = np.zeros((crop[3], crop[2], 4), dtype=np.float32)
    steps = np.zeros((crop[3], crop[2]), dtype=np.uint32)
    nt = threads if threads is not None else max(1, min(16, os.cpu_count() or 1))
    lib().orc_render_slab_stored(_p(v, ctypes.c_uint8), 1 if scalar else 0, _p(dims, ctypes.c_uint32), int(z_store0), int(z_store0 + S),
                                 _p(cam, ctypes.c_float), _p(tfw, ctypes.c_uint32), ctypes.c_float(attenuation), width, height,
                                 int(z_begin), int(z_end), _p(cr, ctypes.c_uint32), _p(out, ctypes.c_float),
                                 _p(steps, ctypes.c_uint32), nt)
    return out, steps


def ssim_u8(img_a, img_b, data_range):
    """skimage.metrics.structural_similarity(a, b, data_range=data_range, channel_axis=-1) for uint8 images, the
    call of scripts/similarity.py:65, restated step by step from scikit-image's published implementation
    (skimage/metrics/_structural_similarity.py, >= 0.19; Wang et al. 2004): float64 copies, scipy's
    uniform_filter of size 7 on X, Y, X*X, Y*Y, X*Y, sample covariance NP/(NP-1), K1 = 0.01, K2 = 0.03, the map
    cropped by (7-1)//2 and averaged per channel, then the mean over channels.  scikit-image is a dependency of
    the script that is not in the reference tree; PINNED against the library itself since round 3: tests/golden/similarity holds
    what scikit-image 0.18.3 returns for these calls (tools/make_similarity_golden.py), equal here to 1e-12."""
    from scipy.ndimage import uniform_filter
    a = np.asarray(img_a, dtype=np.uint8)
    b = np.asarray(img_b, dtype=np.uint8)
    assert a.shape == b.shape and a.ndim in (2, 3)
    if a.ndim == 2:
        a, b = a[:, :, None], b[:, :, None]
    win, k1, k2 = 7, 0.01, 0.03
    assert min(a.shape[:2]) >= win, "win_size exceeds image extent"
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    pad = (win - 1) // 2
    per_channel = []
    for ch in range(a.shape[2]):
        x, y = a[:, :, ch].astype(np.float64), b[:, :, ch].astype(np.float64)
        ux, uy = uniform_filter(x, size=win), uniform_filter(y, size=win)
        uxx, uyy, uxy = uniform_filter(x * x, size=win), uniform_filter(y * y, size=win), uniform_filter(x * y, size=win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
        per_channel.append(s[pad:-pad, pad:-pad].mean(dtype=np.float64))
    return float(np.mean(per_channel))
