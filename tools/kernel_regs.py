#!/usr/bin/env python3
"""Register / LDS use per kernel of pfq_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage).  Usage: tools/kernel_regs.py [filter]"""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "phagefilter_amd", "csrc", "pfq_kernels.hip")
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-c", src,
                      "-o", "/tmp/pfq_regs.o", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+(?:\[[^\]]*\])?): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for name, r in rows.items():
    if flt in name:
        short = re.sub(r"\(.*", "", name).replace("pfq::", "")
        print(f"{short:60s} VGPR {r.get('VGPRs', -1):4d} AGPR {r.get('AGPRs', 0):3d} spillV {r.get('VGPRs Spill', 0):3d} spillS {r.get('SGPRs Spill', 0):4d} "
              f"occ {r.get('Occupancy [waves/SIMD]', -1)} LDS {r.get('LDS Size [bytes/block]', -1)}")
