#!/usr/bin/env python3
"""Per-call-site durations of the LoRA skinny kernels from a rocprofv3 kernel trace (calls repeat with period 8 per layer
in launch order: forward xa for q|k|v, o, gate|up, down; backward dyB / TN products in reverse).
usage: lora_kernel_breakdown.py <kernel_trace.csv>"""
import csv, sys, collections
rows = collections.defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r["Kernel_Name"]
        for key in ("lora_skinny_nt_kernel", "lora_tn_mfma_kernel"):
            if key in n:
                rows[key].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
for key, v in rows.items():
    v.sort()
    per = collections.defaultdict(list)
    for i, (_, d, g) in enumerate(v):
        per[(i % 8, g)].append(d / 1e3)
    print(key, len(v), "launches")
    for (i, g), ds in sorted(per.items()):
        ds.sort()
        print(f"  slot {i} grid {g:>8}: n={len(ds):4d} median {ds[len(ds)//2]:6.1f} us  min {ds[0]:6.1f}")
