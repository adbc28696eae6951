#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of fwgpu_kernels.hip, from the compiler's own report
(-Rpass-analysis=kernel-resource-usage).  usage: python scripts/kernel_resources.py [filter]"""
import os, re, subprocess, sys

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "firewheel_amd", "csrc")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", "-o", "/dev/null", "-x", "hip",
       "fwgpu_kernels.hip", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, cwd=here, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: (?:\S+ )?\s*Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name.replace("void ", "").replace("fwgpu::", ""))}
        rows.append(cur)
        continue
    for key, pat in (("sgpr", r" SGPRs: (\d+)"), ("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
print("%-34s %5s %5s %5s %8s %4s %7s" % ("kernel", "sgpr", "vgpr", "agpr", "scratch", "occ", "lds"))
for r in rows:
    if flt in r["name"]:
        print("%-34s %5d %5d %5d %8d %4d %7d" % (r["name"][:34], r.get("sgpr", -1), r.get("vgpr", -1), r.get("agpr", -1), r.get("scratch", -1),
                                                 r.get("occ", -1), r.get("lds", -1)))
if "error" in err:
    print(err[-2000:])
