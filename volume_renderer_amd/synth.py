"""Seeded synthetic volumes for tests and bench.py: no reference data files exist.

`blobs`: sum of K Gaussian blobs (centres / widths from numpy default_rng(seed)), normalised to [0,1]
and thresholded so that about 65 % of the voxels are exactly 0.  Evaluated separably (per-axis 1-D
Gaussians), slab by slab, so a 512^3 volume needs no multi-GiB temporaries.
"""
import numpy as np

DEFAULT_SEED = 1234
N_BLOBS = 32
ZERO_FRACTION = 0.65


def _blob_params(seed):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(0.15, 0.85, size=(N_BLOBS, 3)).astype(np.float64)
    sigma = rng.uniform(0.04, 0.12, size=(N_BLOBS,)).astype(np.float64)
    weight = rng.uniform(0.5, 1.0, size=(N_BLOBS,)).astype(np.float64)
    return centres, sigma, weight


def _axis_gauss(n, c, s):
    t = (np.arange(n, dtype=np.float64) + 0.5) / n
    return np.exp(-0.5 * ((t[None, :] - c[:, None]) / s[:, None]) ** 2)  # (K, n)


def _field(dims, seed, z0=0, z1=None):
    X, Y, Z = dims
    z1 = Z if z1 is None else z1
    c, s, w = _blob_params(seed)
    gx, gy, gz = _axis_gauss(X, c[:, 0], s), _axis_gauss(Y, c[:, 1], s), _axis_gauss(Z, c[:, 2], s)
    gz = gz[:, z0:z1] * w[:, None]
    # sum_k gz[k,z] * gy[k,y] * gx[k,x]
    yx = (gy[:, :, None] * gx[:, None, :]).reshape(N_BLOBS, Y * X).astype(np.float32)
    return (gz.T.astype(np.float32) @ yx).reshape(z1 - z0, Y, X)


def _levels(seed):
    """threshold and maximum of the field, estimated once at 64^3 so every resolution shares them."""
    f = _field((64, 64, 64), seed)
    return float(np.quantile(f, ZERO_FRACTION)), float(f.max())


def blobs_density(dims, seed=DEFAULT_SEED, slab=32):
    """float32 (Z,Y,X) density in [0,1], ~65 % exact zeros."""
    X, Y, Z = dims
    thr, mx = _levels(seed)
    out = np.empty((Z, Y, X), dtype=np.float32)
    for z0 in range(0, Z, slab):
        z1 = min(Z, z0 + slab)
        f = _field(dims, seed, z0, z1)
        out[z0:z1] = np.clip((f - thr) / (mx - thr), 0.0, 1.0)
    return out


def blobs_scalar_u8_slices(dims, z0, z1, seed=DEFAULT_SEED, slab=32):
    """Slices [z0, z1) of blobs_scalar_u8(dims): what one rank of a slab-partitioned render stores (no whole-volume temporary)."""
    X, Y, Z = dims
    thr, mx = _levels(seed)
    out = np.empty((z1 - z0, Y, X), dtype=np.uint8)
    for a in range(z0, z1, slab):
        b = min(z1, a + slab)
        f = _field(dims, seed, a, b)
        out[a - z0:b - z0] = (np.clip((f - thr) / (mx - thr), 0.0, 1.0) * 255.0).astype(np.uint8)
    return out


def blobs_scalar_u8(dims, seed=DEFAULT_SEED):
    """DAT-style UCHAR volume: uint8 (Z,Y,X)."""
    d = blobs_density(dims, seed)
    return (d * 255.0).astype(np.uint8)


def blobs_rgba_f32(dims, seed=DEFAULT_SEED):
    """NetCDF / DiffDVR-style f32 color(c=4,z,y,x): rgb = a colour ramp of the density, a = density."""
    d = blobs_density(dims, seed)
    out = np.empty((4,) + d.shape, dtype=np.float32)
    out[0] = np.clip(1.5 * d, 0.0, 1.0)
    out[1] = np.clip(1.5 - np.abs(3.0 * d - 1.5), 0.0, 1.0) * (d > 0)
    out[2] = np.clip(1.0 - 1.5 * d, 0.0, 1.0) * (d > 0)
    out[3] = d
    return out


def quantize_rgba8(data_czyx):
    """loaders/netcdf.rs:54-57 in numpy: (v*255.0) as u8 -> uint8 (Z,Y,X,4).  Used to build inputs."""
    v = np.asarray(data_czyx, dtype=np.float32) * np.float32(255.0)
    v = np.nan_to_num(v, nan=0.0)
    q = np.clip(v, 0.0, 255.0).astype(np.uint8)  # astype truncates toward zero
    return np.ascontiguousarray(np.moveaxis(q, 0, -1))


def solid_rgba8(dims, value):
    """every voxel = (value, value, value, value)"""
    X, Y, Z = dims
    return np.full((Z, Y, X, 4), value, dtype=np.uint8)


def random_rgba8(dims, seed=0):
    X, Y, Z = dims
    return np.random.default_rng(seed).integers(0, 256, size=(Z, Y, X, 4), dtype=np.uint8)
