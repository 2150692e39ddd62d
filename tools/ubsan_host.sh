#!/bin/bash
# Host code of libvrm.so under UndefinedBehaviorSanitizer (trap mode: no runtime to preload; UB = SIGILL), device code as built:
#   tools/ubsan_host.sh            builds volume_renderer_amd/variants/libvrm_ubsan.so (here, on the CPU)
#   VRM_LIB=volume_renderer_amd/variants/libvrm_ubsan.so python -m pytest tests -m "not gpu"      (and the -m gpu API tests on the box)
# Round 3: the CPU suite (file parsers incl. the hostile NetCDF/HDF5 files) and the GPU API / eval / sort-last / no-alloc tests ran clean.
set -e
cd "$(dirname "$0")/../volume_renderer_amd/csrc"
make -s
T=$(mktemp -d)
for f in vrm_hdf5 vrm_netcdf vrm_host vrm_eval vrm_api; do
  /opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -fvisibility=hidden \
      -Xarch_host -fsanitize=undefined -Xarch_host -fsanitize-trap=undefined -ffp-contract=off -x hip -c $f.cpp -o $T/$f.o
done
mkdir -p ../variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 vrm_kernels.o $T/*.o -shared -Wl,-rpath,/opt/rocm/lib -lz -lpthread -o ../variants/libvrm_ubsan.so
rm -rf $T
ls -la ../variants/libvrm_ubsan.so
