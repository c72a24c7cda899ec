#!/usr/bin/env python3
"""GPU idle time between kernels of the timed steps from a rocprofv3 kernel trace: span, busy time, gaps by size class.
usage: trace_gaps.py <kernel_trace.csv> [skip_first_n_kernels]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]))
rows.sort()
# keep the last 45 % of the kernels: the steady-state steps (init and warmup are at the front)
n = len(rows)
rows = rows[int(n * 0.55):]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = [max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1)]
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {sum(gaps) / 1e6:.2f} ms ({100 * sum(gaps) / span:.1f} %)")
for lo, hi in ((0, 1000), (1000, 3000), (3000, 10000), (10000, 100000), (100000, 10**12)):
    g = [x for x in gaps if lo <= x < hi]
    print(f"  gaps {lo / 1e3:>6.0f}-{hi / 1e3:<8.0f} us: n={len(g):5d}  total {sum(g) / 1e6:7.3f} ms")
