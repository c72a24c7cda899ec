#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV into a short, committed summary (profiles/).

usage: summarize_rocprof.py <kernel_stats.csv> <out.md> [--steps N] [--note "..."]
Kernel names are shortened to the function name + template arguments; all gemm_nt_kernel<...> instantiations
are also aggregated into one row, which is the number bench.py's `roofline` leg must agree with."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^>(]*>)?)", name)
    s = m.group(1) if m else name[:60]
    if s.startswith("at::native") or "at::native" in name[:40]:
        inner = re.search(r"(normal_kernel|direct_copy_kernel|bfloat16_copy_kernel|FillFunctor|MulFunctor|CUDAFunctor\w*add\w*|\w+Functor\w*)", name)
        s = "torch:" + (inner.group(1) if inner else "elementwise")
    return s


def main():
    src, dst = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 1
    note = sys.argv[sys.argv.index("--note") + 1] if "--note" in sys.argv else ""
    rows = {}
    with open(src) as f:
        for r in csv.DictReader(f):
            k = short(r["Name"])
            c, t = int(r["Calls"]), float(r["TotalDurationNs"])
            a = rows.setdefault(k, [0, 0.0])
            a[0] += c; a[1] += t
    total = sum(v[1] for v in rows.values())
    gemm = [v for k, v in rows.items() if k.startswith("gemm_nt_kernel")]
    gc, gt = sum(v[0] for v in gemm), sum(v[1] for v in gemm)
    out = [f"# rocprofv3 --kernel-trace --stats summary ({src.split('/')[-1]})", "", note, "",
           f"steps profiled: {steps}; total kernel time {total / 1e6:.2f} ms = {total / 1e6 / steps:.2f} ms/step", "",
           f"**gemm_nt_kernel<*> aggregate: {gc} launches, {gt / 1e6:.2f} ms total, average {gt / max(gc, 1) / 1e3:.2f} us/launch, "
           f"{100 * gt / total:.1f} % of kernel time**", "",
           "| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for k, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        if t / total < 0.0005:
            continue
        out.append(f"| {k} | {c} | {t / 1e6:.3f} | {t / c / 1e3:.2f} | {100 * t / total:.2f} |")
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:14]))


if __name__ == "__main__":
    main()
