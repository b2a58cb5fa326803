#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py codename-rvc-fork-3_amd/csrc/conv.hip [name-filter-regex]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", src, "-I", os.path.join(ROOT, "include"),
       "-I", os.path.join(ROOT, "codename-rvc-fork-3_amd", "csrc"), "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: *(.*?) *\[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void rvc::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print(f"{'kernel':58s} {'VGPR':>5s} {'AGPR':>5s} {'spill':>5s} {'scratch':>7s} {'occ':>3s} {'LDS':>7s}")
for r in rows:
    if flt and not flt.search(r["name"]): continue
    print(f"{r['name'][:58]:58s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>5s} "
          f"{r.get('ScratchSize [bytes/lane]','?'):>7s} {r.get('Occupancy [waves/SIMD]','?'):>3s} {r.get('LDS Size [bytes/block]','?'):>7s}")
