#!/bin/bash
# Builds the two timing variants of libvrm.so used for the cost split in DESIGN.md 4.3 (run here, on the CPU):
#   volume_renderer_amd/timing_NO_STEPS.so  chunk prologue + staging + dt advance, no samples taken
#   volume_renderer_amd/timing_NO_STAGE.so  chunk prologue + dt advance only
# On the GPU box copy one over volume_renderer_amd/libvrm.so (the box works on a scratch copy of the tree) and run
# tools/prof_workload.py; total - NO_STEPS = steps, NO_STEPS - NO_STAGE = staging.  The images are wrong by design.
set -e
cd "$(dirname "$0")/../volume_renderer_amd/csrc"
make -s
for f in NO_STAGE NO_STEPS; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc \
      -fvisibility=hidden -DVRM_TIMING_$f -c vrm_kernels.hip -o /tmp/vrm_kernels_$f.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/vrm_kernels_$f.o vrm_api.o vrm_host.o vrm_eval.o vrm_netcdf.o vrm_hdf5.o -shared \
      -Wl,-rpath,/opt/rocm/lib -lz -lpthread -o ../timing_$f.so
  rm -f /tmp/vrm_kernels_$f.o
done
ls -la ../timing_*.so
