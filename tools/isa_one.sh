#!/bin/bash
# One instantiation of the march kernel, device code only, in seconds (run here, on the CPU):
#   tools/isa_one.sh "true,false,false,true,2560,true,false,false" /tmp/isa/sc.s [extra flags]
# template arguments: ALPHA_ONLY, USE_TF, DIAG, SKIP, CAP, PACKED, ERT, FAR
set -e
ARGS=$1; OUT=$2; shift 2
mkdir -p "$(dirname "$OUT")"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-rdc -fvisibility=hidden \
    -I"$(dirname "$0")/../include" "-DVRM_ISA_ONE=$ARGS" "$@" --cuda-device-only -S "$(dirname "$0")/../volume_renderer_amd/csrc/vrm_kernels.hip" -o "$OUT"
grep -E "^\s*; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize)" "$OUT" | head -5
