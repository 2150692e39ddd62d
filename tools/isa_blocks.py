#!/usr/bin/env python3
"""Static instruction counts per basic block of one march_lds_kernel instantiation (run here, on the CPU):
    python tools/isa_blocks.py [mangled-template-fragment]      default: the C3 production kernel
compiles vrm_kernels.hip with --save-temps into /tmp/isa and prints, per block, VALU / SALU / LDS / global counts and the
branch targets: how the chunk prologue's VALU count in DESIGN.md 4.3 was obtained.  ISA_FLAGS="-DVRM_...=..." adds compiler flags."""
import re, subprocess, sys, os
frag = sys.argv[1] if len(sys.argv) > 1 else "ILb0ELb0ELb0ELb1ELi2560ELb0ELb0ELb0E"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs("/tmp/isa", exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
                       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-rdc", "-fvisibility=hidden", "-I" + root + "/include",
                       "--save-temps", *os.environ.get("ISA_FLAGS", "").split(), "-c", root + "/volume_renderer_amd/csrc/vrm_kernels.hip", "-o", "k.o"], cwd="/tmp/isa",
                      stderr=subprocess.DEVNULL)
lines = open("/tmp/isa/vrm_kernels-hip-amdgcn-amd-amdhsa-gfx950.s").read().split("\n")
name = frag if "kernel" in frag else "march_lds_kernel" + frag   # e.g. march_wg_kernelILb0ELb0ELb1ELi6144ELb0ELb0E
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().split(";")[0].rstrip().endswith(":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
open("/tmp/isa/sel.s", "w").write("\n".join(lines[start:end + 1]))
blocks, cur = [], ["entry", 0, 0, 0, 0, []]
for ln in lines[start:end + 1]:
    s = ln.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        blocks.append(cur)
        cur = [m.group(1), 0, 0, 0, 0, []]
        continue
    if not s or s[0] in ";.":
        continue
    op = s.split()[0]
    if op.startswith("v_"): cur[1] += 1
    elif op.startswith("s_"): cur[2] += 1
    elif op.startswith("ds_"): cur[3] += 1
    elif op.split("_")[0] in ("global", "buffer", "flat", "scratch"): cur[4] += 1
    if op.startswith("s_cbranch") or op == "s_branch": cur[5].append(s.split()[-1])
blocks.append(cur)
for b in blocks:
    print(f"{b[0]:12s} V {b[1]:4d} S {b[2]:4d} DS {b[3]:2d} G {b[4]:2d} -> {','.join(b[5])}")
meta = [l.strip() for l in lines[end:end + 80] if re.search(r"NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize", l)]
print("; ".join(meta))
print("total V", sum(b[1] for b in blocks), "S", sum(b[2] for b in blocks), "; listing in /tmp/isa/sel.s")
