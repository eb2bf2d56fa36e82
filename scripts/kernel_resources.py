#!/usr/bin/env python3
"""Table of per-kernel resource usage from a `build.py --force -v` log (hipcc -Rpass-analysis=kernel-resource-usage):
name, VGPRs, AGPRs, scratch bytes/lane (spills), occupancy, LDS bytes.   python scripts/kernel_resources.py build.log [filter]"""
import re
import subprocess
import sys

rows, cur = [], None
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass-analysis", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
names = [r["name"] for r in rows]
try:
    dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
except Exception:
    dem = names
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d)
    if flt and flt not in d:
        continue
    print(f"{d[:90]:90s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>5} "
          f"occ {r.get('Occupancy [waves/SIMD]','?'):>2} lds {r.get('LDS Size [bytes/block]','?'):>6} sgpr {r.get('SGPRs','?'):>4}")
