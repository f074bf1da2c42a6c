#!/usr/bin/env python3
"""Per-stream timeline of one training step from a rocprofv3 kernel trace (development tool).
    python tools/timeline.py gpurun_out/prof_x/r_kernel_trace.csv [step_from_end=3]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r['Stream_Id']) for r in rows)
ad = [i for i, e in enumerate(ev) if 'opt_adamw' in e[2]]
i0, i1 = ad[-k - 1] + 1, ad[-k] + 1
t0 = ev[i0][0]
def short(n):
    n = re.sub(r'_ZN4ge2e\d+', '', n).replace('ge2e::', '').replace('void ', '')
    return n[:44]
last = {}
gaps = {}
for s, e, n, q in ev[i0:i1]:
    gap = (s - last.get(q, s)) / 1e3
    gaps[q] = gaps.get(q, 0) + max(gap, 0)
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} gap {gap:7.1f} q{q} {short(n)}")
    last[q] = e
print('span', (ev[i1 - 1][1] - t0) / 1e3, 'gaps per stream', gaps)
