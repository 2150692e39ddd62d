"""Writes tests/golden/*.npz: seeded inputs + the CPU oracle's outputs for them.

The reference ships no golden vectors for this path and cannot be run, so these
are produced by oracle/vr_oracle.c and say "parity unpinned" in their metadata; they freeze the
oracle's behaviour (tests/test_golden.py re-derives them on CPU) and give the GPU tests vectors that
do not depend on the oracle being rebuilt on the GPU box.

usage: python tools/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from volume_renderer_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# name, volume builder, (ix, iy, xdiv, ydiv, radius), (W, H), attenuation, use_tf
CASES = [
    ("blobs_rgba_default", lambda: synth.quantize_rgba8(synth.blobs_rgba_f32((32, 32, 32))), (1, 3, 8, 8, 2.5), (64, 64), 1.0, False),
    ("blobs_rgba_tf_att20", lambda: synth.quantize_rgba8(synth.blobs_rgba_f32((32, 32, 32))), (5, 6, 8, 8, 2.5), (64, 64), 20.0, True),
    ("blobs_scalar_dat", lambda: np.repeat(synth.blobs_scalar_u8((32, 32, 32))[..., None], 4, 3), (0, 4, 8, 8, 2.5), (64, 64), 1.0, False),
    ("random_noncubic_att100", lambda: synth.random_rgba8((20, 9, 33), 11), (11, 5, 16, 17, 2.5), (56, 40), 100.0, False),
    ("default_grid_3x3x3", lambda: np.tile(np.array([0, 0, 0, 255], np.uint8), (3, 3, 3, 1)), (3, 2, 8, 8, 2.5), (48, 48), 1.0, False),
    ("radius100_reference_default", lambda: synth.random_rgba8((8, 8, 8), 2), (2, 5, 8, 8, 100.0), (96, 96), 5.0, True),
]


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, build, (ix, iy, xd, yd, rad), (W, H), att, tf in CASES:
        vox = build()
        cam = oracle.camera_uniform(oracle.sphere_position(ix, iy, xd, yd, rad), aspect=W / H)
        r = oracle.render(vox, cam, W, H, attenuation=att, tf=oracle.tf_words(tf), want_steps=True)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), voxels=vox, camera=cam, width=W, height=H,
                            attenuation=np.float32(att), use_tf=tf, rgba8=r["rgba8"], f32=r["f32"], steps=r["steps"],
                            provenance="oracle/vr_oracle.c via tools/make_golden.py; parity unpinned (no reference fixtures exist)")
        print(name, vox.shape, (W, H), "samples", r["samples"], "nonzero px", int((r["rgba8"][..., 3] > 0).sum()))


if __name__ == "__main__":
    main()
