"""Sort-last rendering of a volume partitioned into z slabs over the ranks of a process group.

The reference keeps the whole volume in one 3D texture (texture_3d.rs:22-80) and has nothing to partition; this is
the scale-out form of the same ray march for volumes beyond one GPU's share:

* rank r owns the slices [z_r, z_{r+1}) (`slab_ranges`) plus one ghost slice either side (`VoxelGrid.new_slab`);
* every rank marches the SAME global sample lattice (box, t_min, running dt sum of the whole volume) and blends only
  the samples whose position lies in its half-open slab -> a premultiplied f32 RGBA partial per pixel
  (`RayMarcher.draw_partial`);
* ONE exchange step, direct-send: the image is cut into `world` row bands, an all-to-all moves band b of every
  rank's partial to rank b (RCCL `all_to_all_single` on device tensors; W*H/G pixels * 16 B per pair and view);
* rank b composites the G partials of its band with the associative "over" operator in the order each pixel's ray
  visits the slabs (ascending z iff rd.z >= 0) and stores RGBA8 like the single-GPU path (`RayMarcher.composite`).

"Over" is not a reduce op RCCL knows, hence all-to-all + local compose rather than an all-reduce.  torch is used for
device memory and the collective only.
"""
import numpy as np


def slab_ranges(z, n):
    """Balanced contiguous slice ranges [(z0, z1)] * n of z slices (the first z % n slabs get one slice more)."""
    z, n = int(z), int(n)
    if not 0 < n <= z:
        raise ValueError(f"cannot cut {z} slices into {n} slabs")
    q, r = divmod(z, n)
    out, a = [], 0
    for i in range(n):
        b = a + q + (1 if i < r else 0)
        out.append((a, b))
        a = b
    return out


def stored_range(z, z0, z1):
    """Slices a slab context stores: the slab plus one ghost slice either side, clipped to the volume."""
    return max(z0 - 1, 0), min(z1 + 1, int(z))


def row_bands(height, n):
    """Row bands [(row0, rows)] * n of the image for the direct-send exchange (may be empty for n > height)."""
    q, r = divmod(int(height), int(n))
    out, a = [], 0
    for i in range(n):
        k = q + (1 if i < r else 0)
        out.append((a, k))
        a += k
    return out


def exchange_bands(partial, group=None):
    """The one exchange step.  partial: torch tensor (n_views, H, W, 4) f32, this rank's slab partial (CPU tensor with
    gloo, device tensor with RCCL).  Returns (stack, row0, rows): stack (world, n_views, rows, W, 4) = band `rank` of
    every rank's partial, indexed by source rank = slab index."""
    import torch
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    n, H, W, C = partial.shape
    bands = row_bands(H, world)
    row0, rows = bands[rank]
    if world == 1:
        return partial.reshape(1, n, H, W, C), 0, H
    # send buffer: band-major so that each destination's part is contiguous: [band][view][row][x][c]
    send = torch.cat([partial[:, a:a + k].reshape(-1) for a, k in bands])
    in_split = [n * k * W * C for _, k in bands]
    out_split = [n * rows * W * C] * world
    recv = torch.empty(sum(out_split), dtype=partial.dtype, device=partial.device)
    dist.all_to_all_single(recv, send, output_split_sizes=out_split, input_split_sizes=in_split, group=group)
    return recv.reshape(world, n, rows, W, C), row0, rows


def composite_numpy(stack, ascending):
    """CPU restatement of the compositing kernel (tests, gloo rehearsal): stack (G, ..., 4) f32 partials, slab 0 lowest
    in z; ascending: bool array (...) = rd.z >= 0 per pixel.  f32 arithmetic with the kernel's operation order
    (c + (1-alpha)*p as one FMA is emulated in f64 and rounded once)."""
    stack = np.asarray(stack, dtype=np.float32)
    G = stack.shape[0]
    asc = np.asarray(ascending, dtype=bool)
    acc = np.zeros(stack.shape[1:], dtype=np.float32)
    for s in range(G):
        p = np.where(asc[..., None], stack[s], stack[G - 1 - s])
        om = (np.float32(1.0) - acc[..., 3:4]).astype(np.float32)
        acc = (om.astype(np.float64) * p.astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    return acc


def ascending_mask(cam36, width, height, row0=0, rows=None):
    """rd.z >= 0 per pixel as fs_main forms the ray (wgsl:98-108): the sign of the un-normalised z."""
    rows = height - row0 if rows is None else rows
    c = np.asarray(cam36, dtype=np.float32)
    ivp = c[20:36]
    px = (np.arange(width, dtype=np.float32) + np.float32(0.5)) / np.float32(width)
    py = np.float32(1.0) - (np.arange(row0, row0 + rows, dtype=np.float32) + np.float32(0.5)) / np.float32(height)
    sx = (px * np.float32(2.0) - np.float32(1.0))[None, :]
    sy = (py * np.float32(2.0) - np.float32(1.0))[:, None]
    wz = ((ivp[2] * sx + ivp[6] * sy) + ivp[10]) + ivp[14]
    return wz >= 0


def unorm8(v):
    """Rgba8Unorm store (mod.rs:79-83): round(clamp(v,0,1)*255)."""
    v = np.clip(np.nan_to_num(np.asarray(v, dtype=np.float32), nan=0.0), 0.0, 1.0)
    return np.floor(v * np.float32(255.0) + np.float32(0.5)).astype(np.uint8)


class SortLastRenderer:
    """One rank of a sort-last render: holds this rank's slab in a RayMarcher and renders views cooperatively."""

    def __init__(self, ray_marcher, dims, group=None):
        import torch.distributed as dist
        self.rm = ray_marcher
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        self.dims = tuple(int(d) for d in dims)
        self.z0, self.z1 = slab_ranges(self.dims[2], self.world)[self.rank]
        self.s0, self.s1 = self.rm.voxel_grid.new_slab(self.dims, self.z0, self.z1)

    def upload_rgba8(self, stored_slices_zyxc):
        """voxels of the stored slices [s0, s1): uint8 (s1-s0, Y, X, 4)."""
        v = np.ascontiguousarray(stored_slices_zyxc, dtype=np.uint8)
        if v.shape != (self.s1 - self.s0, self.dims[1], self.dims[0], 4):
            raise ValueError(f"expected slices [{self.s0},{self.s1}) as {(self.s1 - self.s0, self.dims[1], self.dims[0], 4)}, got {v.shape}")
        self.rm.voxel_grid.upload_rgba8(v)

    def upload_scalar_u8(self, stored_slices_zyx):
        """DAT-style density of the stored slices [s0, s1): uint8 (s1-s0, Y, X), replicated to RGBA on the GPU (dat.rs:74)."""
        v = np.ascontiguousarray(stored_slices_zyx, dtype=np.uint8)
        if v.shape != (self.s1 - self.s0, self.dims[1], self.dims[0]):
            raise ValueError(f"expected slices [{self.s0},{self.s1}) as {(self.s1 - self.s0, self.dims[1], self.dims[0])}, got {v.shape}")
        self.rm.voxel_grid.upload_scalar(v.reshape(-1), (self.dims[0], self.dims[1], self.s1 - self.s0), 1)

    def render(self, camera_uniforms, width, height, device, host_exchange=False):
        """Render the views cooperatively.  Returns (band_rgba8, row0, rows): this rank's row band of every view as a
        torch uint8 tensor (n, rows, W, 4) on `device`.  host_exchange: route the all-to-all through host tensors (a
        gloo process group, e.g. several ranks rehearsing on one GPU); the default hands device tensors to RCCL."""
        import torch
        n = len(camera_uniforms)
        partial = torch.empty((n, height, width, 4), dtype=torch.float32, device=device)
        self.rm.draw_partial(camera_uniforms, width, height, partial.data_ptr())
        self.rm.synchronize()                       # the library's stream -> visible to torch / RCCL
        if host_exchange:
            stack, row0, rows = exchange_bands(partial.cpu(), self.group)
            stack = stack.to(device)
        else:
            stack, row0, rows = exchange_bands(partial, self.group)
        out = torch.empty((n, rows, width, 4), dtype=torch.uint8, device=device)
        if rows > 0:
            per_view = stack.permute(1, 0, 2, 3, 4).contiguous()          # (n, G, rows, W, 4)
            torch.cuda.synchronize(device)          # torch's stream -> visible to the library's stream
            for i, cam in enumerate(camera_uniforms):
                self.rm.composite(cam, per_view[i].data_ptr(), self.world, width, height, row0, rows, out[i].data_ptr(), None)
            self.rm.synchronize()
        return out, row0, rows
