#!/usr/bin/env python3
"""Register / LDS / spill figures of the kernels in a compiled unit (from the code object's metadata):
    python scripts/kres.py pypmc_amd/csrc/build/pmc_fused_d8_p0.o [name-filter]
"""
import re
import subprocess
import sys
import tempfile
import os

obj = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
with tempfile.TemporaryDirectory() as tmp:
    # a host object bundles the device code object: unbundle it
    out = os.path.join(tmp, "dev.co")
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + obj,
                        "--targets=hip-amdgcn-amd-amdhsa--gfx950", "--output=" + out], capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(out) or os.path.getsize(out) == 0:
        # -fgpu-rdc-less objects keep the fat binary in a section
        fat = os.path.join(tmp, "fat.bin")
        subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + out])
    notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", out], capture_output=True, text=True).stdout
for blk in notes.split("- .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if flt not in name:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    g = lambda key: int(re.search(r"\." + key + r":\s+(\d+)", blk).group(1))
    agpr = int(blk.split()[0])
    print("%-60s vgpr %3d agpr %3d sgpr %3d lds %6d spill(v) %3d scratch %4d" % (
        dem[:60], g("vgpr_count"), agpr, g("sgpr_count"), g("group_segment_fixed_size"), g("vgpr_spill_count"),
        g("private_segment_fixed_size")))
