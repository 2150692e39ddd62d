"""PNG-inclusive rate of the file-writing exporter (vrm_export_sphere: march + read-back + PNG encode + cameras.json),
reported beside bench.py's views/s as SURVEY.md 8d asks.  usage: python tools/export_bench.py [dim] [res] [xdiv] [ydiv]"""
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import volume_renderer_amd as vr  # noqa: E402
from volume_renderer_amd import synth  # noqa: E402

D = int(sys.argv[1]) if len(sys.argv) > 1 else 512
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
xdiv = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ydiv = int(sys.argv[4]) if len(sys.argv) > 4 else 17
rm = vr.RayMarcher(0, R, R)
rm.voxel_grid.upload_rgba8(synth.quantize_rgba8(synth.blobs_rgba_f32((D, D, D))))
out = tempfile.mkdtemp(prefix="vrm_export_", dir="/tmp")
try:
    rm.export_sphere(xdiv, 3, 2.5, out)                      # warm-up
    t0 = time.perf_counter()
    rm.export_sphere(xdiv, ydiv, 2.5, out)
    dt = time.perf_counter() - t0
    n = xdiv * (ydiv - 1)
    size = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    print(f"export_sphere: {n} views {R}^2 of {D}^3 -> PNG + cameras.json in {dt:.2f} s = {n / dt:.1f} views/s "
          f"({size / 1e6:.0f} MB written, {os.cpu_count()} host CPUs visible)")
finally:
    shutil.rmtree(out, ignore_errors=True)
    rm.close()
