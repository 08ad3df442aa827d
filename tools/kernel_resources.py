#!/usr/bin/env python3
"""Compiles one .hip file of the package with -Rpass-analysis=kernel-resource-usage and prints one line per kernel:
registers, spills, scratch, occupancy, LDS.  usage: tools/kernel_resources.py csrc/fdr_panel.hip [filter] [extra flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "parallel-implementation-of-frequency-domain-image-restoration-using-fft_amd")
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
       "-fPIC", "-c", os.path.join(PKG, src), "-o", "/tmp/_kr.o", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    short = re.sub(r"\(.*", "", r["name"]).replace("void fdr::", "")
    if flt in short:
        print("%-52s vgpr %3d agpr %3d spill %3d scratch %4d occ %d lds %6d sgpr %3d" % (short[:52], r.get("vgpr", -1), r.get("agpr", 0), r.get("vspill", -1),
              r.get("scratch", -1), r.get("occ", -1), r.get("lds", -1), r.get("sgpr", -1)))
