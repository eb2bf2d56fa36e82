#!/usr/bin/env python3
"""Timeline of ONE steady-state sweep from a rocprofv3 --kernel-trace CSV: per kernel start offset, duration and the gap to
the previous kernel's end (microseconds), plus the sums.   python scripts/sweep_timeline.py <kernel_trace.csv> [sweep_index_from_end]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"^vbmf::", "", n)
    m = re.match(r"(\w+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or ""))[:60] if m else n[:60]


# a sweep starts at a launch of the streaming kernel that follows a pair/gram reduce (pass 1); find pass-1 launches
idx = [i for i, e in enumerate(ev) if "stream_gemm_kernel" in e[2] or "stream_lds8_kernel" in e[2]]
# pass 1 and pass 2 alternate in the run loop: take launches from the end
starts = idx[::2] if len(idx) % 2 == 0 else idx[1::2]
s0, s1 = starts[-back - 1], starts[-back]
t0 = ev[s0][0]
prev_end = None
tot_k = 0
for a, b, n in ev[s0:s1]:
    gap = (a - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(a - t0) / 1e3:9.1f} us  +{(b - a) / 1e3:8.1f}  gap {gap:6.1f}  {short(n)}")
    tot_k += b - a
    prev_end = b
print(f"sweep: {(ev[s1][0] - t0) / 1e3:.1f} us wall, kernels {tot_k / 1e3:.1f} us, gaps {(ev[s1][0] - t0 - tot_k) / 1e3:.1f} us")
