#!/usr/bin/env python3
"""Per-STEP kernel table from a rocprofv3 --kernel-trace CSV of bench.py: model construction and warm-up are cut off (a step
starts at its logmel_fft_kernel launch; rounds 1-4: logmel_init_kernel / logmel_power_kernel), so the table holds exactly what one training step launches.

usage: summarize_trace_steps.py <kernel_trace.csv> <out.md> [--skip N] [--note "..."]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^>(]*>)?)", name)
    s = m.group(1) if m else name[:60]
    if "at::native" in name[:60] or s.startswith("at::"):
        inner = re.search(r"(normal_kernel|direct_copy_kernel|bfloat16_copy_kernel|FillFunctor|MulFunctor|CUDAFunctor\w*add\w*|\w+Functor\w*)", name)
        s = "torch:" + (inner.group(1) if inner else "elementwise")
    return s


def main():
    src, dst = sys.argv[1], sys.argv[2]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 1
    note = sys.argv[sys.argv.index("--note") + 1] if "--note" in sys.argv else ""
    rows = []
    with open(src) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "logmel_init_kernel" in r[2] or "logmel_power_kernel" in r[2]]
    if not starts:                                         # round 5: the init launch is gone; a step's first kernel is the FFT kernel
        starts = [i for i, r in enumerate(rows) if "logmel_fft_kernel" in r[2]]
    if len(starts) <= skip + 1:
        raise SystemExit(f"only {len(starts)} steps in the trace")
    lo, hi = starts[skip], starts[-1]                      # whole steps only: from step `skip` to the start of the last one
    steps = len(starts) - 1 - skip
    sel = rows[lo:hi]
    agg = {}
    for s, e, n in sel:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1; a[1] += e - s
    total = sum(v[1] for v in agg.values())
    span = sel[-1][1] - sel[0][0]
    gemm = [v for k, v in agg.items() if k.startswith("gemm_nt_kernel")]
    gc, gt = sum(v[0] for v in gemm), sum(v[1] for v in gemm)
    out = [f"# per-step kernel table ({src.split('/')[-1]})", "", note, "",
           f"steps: {steps}; kernel time {total / 1e6 / steps:.3f} ms/step; wall span {span / 1e6 / steps:.3f} ms/step "
           f"(GPU idle between kernels {100 * (1 - total / span):.1f} %); {sum(v[0] for v in agg.values()) // steps} launches/step", "",
           f"**gemm_nt_kernel<*> aggregate: {gc // steps} launches/step, {gt / 1e6 / steps:.3f} ms/step, average {gt / max(gc, 1) / 1e3:.2f} us/launch, "
           f"{100 * gt / total:.1f} % of kernel time**", "",
           "| kernel | calls/step | ms/step | avg us | % |", "|---|---:|---:|---:|---:|"]
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if t / total < 0.0003:
            continue
        out.append(f"| {k} | {c / steps:.1f} | {t / 1e6 / steps:.3f} | {t / c / 1e3:.2f} | {100 * t / total:.2f} |")
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:60]))


if __name__ == "__main__":
    main()
