#!/usr/bin/env python3
"""Kernel resource usage of every HIP translation unit, from the code-object metadata of `hipcc -S` (what
-Rpass-analysis=kernel-resource-usage prints, as a table): python3 tools/isa/resource_usage.py > profiles/rNN_kernel_resource_usage.txt"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "poreover_amd"))
import build as _build   # (the objects of the library and each one's options)
SRC = sorted(_build.OBJECTS)
print("# kernel resource usage, hipcc -O3 --offload-arch=gfx950 -S (code-object metadata), the sources of this commit")
print("# kernel | VGPRs | AGPRs | SGPRs | SGPR spills | VGPR spills | scratch B/lane | LDS B/block | waves/SIMD (512 / VGPRs, at most 8)")
for oname, sname, extra in SRC:
    src = os.path.join(ROOT, "poreover_amd", "csrc", sname)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "x.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value", "-Wno-unused-function",
                        *extra, "-S", "--cuda-device-only", "-o", out, src], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    meta = txt[txt.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in txt else ""
    print("## " + oname + ((" (" + sname + " " + " ".join(extra) + ")") if extra else ""))
    for blk in meta.split("  - .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "0"])[1]
        name = g("name")
        try:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dem = name
        dem = re.sub(r"\(.*", "", dem).replace("void ", "").replace("(anonymous namespace)::", "")
        agpr = re.match(r"\s*(\d+)", blk).group(1)
        v = int(g("vgpr_count"))
        waves = min(8, 512 // max(((v + 7) // 8) * 8, 8))
        print("%s | %s | %s | %s | %s | %s | %s | %s | %d" % (dem, v, agpr, g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"),
                                                            g("private_segment_fixed_size"), g("group_segment_fixed_size"), waves))
