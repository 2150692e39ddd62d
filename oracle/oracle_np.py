"""numpy twin of oracle/vr_oracle.c -- an independently written restatement of the same reference code
(src/ray_marcher/raymarcher.wgsl:98-299, src/voxel/grid.rs:265-280), vectorised over rays.

TEST INFRASTRUCTURE ONLY.  It exists so that the C oracle is not its own and only witness: the two
are written separately (this one array-at-a-time in numpy float32, the C one ray-at-a-time) and
tests/test_oracle_twin.py requires them to agree.  Small images only (the step loop is in Python).
"""
import numpy as np

F = np.float32
MAX_STEP_AMOUNT = 5000  # wgsl:96
STEP = F(0.001)  # wgsl:126


def grid_box(dims_xyz):
    """VoxelGridUniform::new (grid.rs:265-280): box_size = dims / min(dims), box_min = -box_size/2."""
    d = np.asarray(dims_xyz, dtype=np.uint32)
    size = d.astype(F) / F(d.min())
    return (-size / F(2.0)).astype(F), size.astype(F)


def pixel_rays(cam36, W, H):
    """fs_main wgsl:98-108 for every pixel; framebuffer origin top-left, uv = (ndc+1)/2 at pixel centres."""
    cam = np.asarray(cam36, dtype=F)
    ivp = cam[20:36].reshape(4, 4)  # ivp[col, row]
    px = np.arange(W, dtype=F)[None, :].repeat(H, 0)
    py = np.arange(H, dtype=F)[:, None].repeat(W, 1)
    uvx = (px + F(0.5)) / F(W)
    uvy = F(1.0) - (py + F(0.5)) / F(H)
    sx = uvx * F(2.0) - F(1.0)
    sy = uvy * F(2.0) - F(1.0)
    w = [((ivp[0, r] * sx + ivp[1, r] * sy) + ivp[2, r] * F(1.0)) + ivp[3, r] * F(1.0) for r in range(3)]
    ln = np.sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2])
    rd = np.stack([w[0] / ln, w[1] / ln, w[2] / ln], axis=-1).astype(F)
    ro = cam[0:3]
    return ro, rd


def slab_test(ro, rd, bmin, bsize):
    """aabb_intersect wgsl:247-299, vectorised; returns hit mask, t_min, t_max."""
    bmax = (bmin + bsize).astype(F)
    with np.errstate(divide="ignore", invalid="ignore"):
        lo = [(bmin[a] - ro[a]) / rd[..., a] for a in range(3)]
        hi = [(bmax[a] - ro[a]) / rd[..., a] for a in range(3)]
    for a in range(3):
        sw = lo[a] > hi[a]
        lo[a], hi[a] = np.where(sw, hi[a], lo[a]), np.where(sw, lo[a], hi[a])
    t_min, t_max = lo[0], hi[0]
    hit = ~((t_min > hi[1]) | (lo[1] > t_max))
    t_min = np.where(lo[1] > t_min, lo[1], t_min)
    t_max = np.where(hi[1] < t_max, hi[1], t_max)
    hit &= ~((t_min > hi[2]) | (lo[2] > t_max))
    t_min = np.where(lo[2] > t_min, lo[2], t_min)
    t_max = np.where(hi[2] < t_max, hi[2], t_max)
    return hit, t_min.astype(F), t_max.astype(F)


def sample(vox, p, bmin, bsize):
    """scene wgsl:199-238 + the linear / clamp-to-edge sampler (texture_3d.rs:66-76). vox: (Z,Y,X,4) u8."""
    Z, Y, X, _ = vox.shape
    dims = (X, Y, Z)
    bmax = (bmin + bsize).astype(F)
    inside = np.ones(p.shape[:-1], dtype=bool)
    for a in range(3):
        inside &= ~(p[..., a] < bmin[a])
    for a in range(3):
        inside &= ~(p[..., a] >= bmax[a])
    out = np.zeros(p.shape[:-1] + (4,), dtype=F)
    if not inside.any():
        return out
    q = p[inside]
    lo, hi, fr = [], [], []
    for a in range(3):
        rel = (q[:, a] - bmin[a]) / bsize[a]
        u = rel * F(dims[a]) - F(0.5)
        fl = np.floor(u)
        fr.append((u - fl).astype(F))
        i = fl.astype(np.int64)
        lo.append(np.clip(i, 0, dims[a] - 1))
        hi.append(np.clip(i + 1, 0, dims[a] - 1))
    tex = vox.astype(F) / F(255.0)

    def lerp(a, b, w):
        return a + w[:, None] * (b - a)

    c00 = lerp(tex[lo[2], lo[1], lo[0]], tex[lo[2], lo[1], hi[0]], fr[0])
    c10 = lerp(tex[lo[2], hi[1], lo[0]], tex[lo[2], hi[1], hi[0]], fr[0])
    c01 = lerp(tex[hi[2], lo[1], lo[0]], tex[hi[2], lo[1], hi[0]], fr[0])
    c11 = lerp(tex[hi[2], hi[1], lo[0]], tex[hi[2], hi[1], hi[0]], fr[0])
    c0 = lerp(c00, c10, fr[1])
    c1 = lerp(c01, c11, fr[1])
    out[inside] = lerp(c0, c1, fr[2])
    return out


def transfer(a, col_a, col_b, col_c):
    """wgsl:145-147 with remap (wgsl:240-242) and WGSL mix = e1*(1-e3) + e2*e3."""
    t_ab = (F(0.0) + (a - F(0.0)) * (F(1.0) - F(0.0)) / (F(0.5) - F(0.0)))[..., None]
    t_bc = (F(0.0) + (a - F(0.5)) * (F(1.0) - F(0.0)) / (F(1.0) - F(0.5)))[..., None]
    ab = col_a * (F(1.0) - t_ab) + col_b * t_ab
    bc = col_b * (F(1.0) - t_bc) + col_c * t_bc
    st = (a >= F(0.5)).astype(F)[..., None]
    return ab * (F(1.0) - st) + bc * st


def render(vox, cam36, W, H, attenuation=1.0, use_tf=False, colors=None):
    """raymarch wgsl:119-196 for a whole image. Returns dict(f32 (H,W,4), rgba8, steps (H,W))."""
    vox = np.ascontiguousarray(vox, dtype=np.uint8)
    Z, Y, X, _ = vox.shape
    bmin, bsize = grid_box((X, Y, Z))
    ro, rd = pixel_rays(cam36, W, H)
    hit, t_min, t_max = slab_test(ro, rd, bmin, bsize)
    att = F(attenuation)
    if colors is None:
        colors = ((0.117, 0.188, 0.62), (0.7294, 0.7294, 0.7294), (0.5725, 0.0, 0.039))  # grid.rs:285-287
    col_a, col_b, col_c = (np.asarray(c, dtype=F)[:3] for c in colors)

    color = np.zeros((H, W, 3), dtype=F)
    alpha = np.zeros((H, W), dtype=F)
    steps = np.zeros((H, W), dtype=np.uint32)
    dt = np.where(hit, t_min, F(0.0)).astype(F)
    active = hit.copy()
    for _ in range(MAX_STEP_AMOUNT):
        if not active.any():
            break
        idx = np.nonzero(active)
        steps[idx] += 1
        d = dt[idx]
        p = (ro[None, :] + rd[idx] * d[:, None]).astype(F)
        s = sample(vox, p, bmin, bsize)
        a = s[:, 3]
        alpha_src = F(1.0) - np.exp(-a * STEP * att, dtype=F)
        src = transfer(a, col_a, col_b, col_c) if use_tf else s[:, :3]
        al = alpha[idx]
        color[idx] = color[idx] + ((F(1.0) - al) * alpha_src)[:, None] * src
        al = al + (F(1.0) - al) * alpha_src
        alpha[idx] = al
        done = al >= F(1.0)
        d = np.where(done, d, d + STEP).astype(F)
        dt[idx] = d
        done |= d >= t_max[idx]
        active[idx] = ~done
    f32 = np.concatenate([color, alpha[..., None]], axis=-1).astype(F)
    f32[~hit] = 0
    clamped = np.clip(np.nan_to_num(f32, nan=0.0), 0.0, 1.0)
    rgba8 = np.floor(clamped * F(255.0) + F(0.5)).astype(np.uint8)
    return {"f32": f32, "rgba8": rgba8, "steps": steps}
