#!/bin/bash
# Kernel A/B builds (run here, on the CPU): tools/ab_builds.sh NAME [extra hipcc flags / -D switches ...]
# compiles vrm_kernels.hip with the extra flags and links volume_renderer_amd/variants/libvrm_NAME.so against the other
# objects of the in-tree build.  On the GPU box: VRM_LIB=volume_renderer_amd/variants/libvrm_NAME.so python tools/prof_workload.py ...
set -e
cd "$(dirname "$0")/../volume_renderer_amd/csrc"
NAME=$1; shift
mkdir -p ../variants
make -s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc \
    -fvisibility=hidden "$@" -c vrm_kernels.hip -o /tmp/vrm_kernels_$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 /tmp/vrm_kernels_$NAME.o vrm_api.o vrm_host.o vrm_eval.o vrm_netcdf.o vrm_hdf5.o -shared \
    -Wl,-rpath,/opt/rocm/lib -lz -lpthread -o ../variants/libvrm_$NAME.so
rm -f /tmp/vrm_kernels_$NAME.o
ls -la ../variants/libvrm_$NAME.so
