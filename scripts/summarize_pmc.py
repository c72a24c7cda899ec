#!/usr/bin/env python3
"""Condense the rocprofv3 --pmc passes of scripts/gpu_pmc.sh into profiles/<tag>_pmc_summary.md.
usage: summarize_pmc.py <pmc_dir> <out.md> [--note "..."]
Per (kernel, grid size): mean counter value per launch.  FETCH_SIZE is doubled (gfx950 correction of
MI355X_MICROARCH.md, HBM section: 128-B requests tallied at 64 B for 16-B/lane streaming reads); FETCH/WRITE are KB."""
import csv, glob, os, re, sys, collections

src, dst = sys.argv[1], sys.argv[2]
note = sys.argv[sys.argv.index("--note") + 1] if "--note" in sys.argv else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(src, "*counter_collection.csv"))):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            if not any(t in n for t in ("gemm_nt", "attn_", "lora_", "linear_small", "dec_", "logmel_fft")):
                continue
            m = re.match(r"(?:void )?([\w:]+(?:<[^>(]*>)?)", n)
            key = (m.group(1) if m else n[:50], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
            a = acc[key][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
# launch-weighted aggregate over every gemm_nt launch (bench.py reports it as roofline.traffic)
tot = {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]}
for (k, _g), cs in acc.items():
    if "gemm_nt" in k:
        for c in tot:
            if c in cs:
                tot[c][0] += cs[c][0]; tot[c][1] += cs[c][1]
agg = None
if tot["FETCH_SIZE"][1] and tot["WRITE_SIZE"][1]:
    agg = {"launches": tot["FETCH_SIZE"][1], "fetch_mb_per_launch": 2 * tot["FETCH_SIZE"][0] / tot["FETCH_SIZE"][1] / 1024,
           "write_mb_per_launch": tot["WRITE_SIZE"][0] / tot["WRITE_SIZE"][1] / 1024}
    if "--json" in sys.argv:
        import json
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as jf:
            json.dump(dict(agg, source=os.path.basename(dst), note=note,
                           method="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH x2 (gfx950 correction), KB -> MB"), jf, indent=1)
rows = []
for key, cs in acc.items():
    g = lambda c: cs[c][0] / cs[c][1] if c in cs and cs[c][1] else float("nan")
    busy, mfma = g("SQ_BUSY_CYCLES"), g("SQ_VALU_MFMA_BUSY_CYCLES")
    rows.append((key, g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_LDS_IDX_ACTIVE"), 1), g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
                 g("SQ_WAIT_ANY") / max(g("SQ_WAVE_CYCLES"), 1), g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1),
                 g("SQ_INSTS_VALU") / max(g("SQ_INSTS_MFMA"), 1), g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1),
                 2 * g("FETCH_SIZE") / 1024, g("WRITE_SIZE") / 1024, mfma / max(busy, 1)))
rows.sort(key=lambda r: (r[0][0], -r[0][1]))
with open(dst, "w") as out:
    out.write(f"# PMC summary ({os.path.basename(src.rstrip('/'))}; scripts/gpu_pmc.sh: separate --pmc passes, kernel-trace only)\n\n{note}\n\n"
              "FETCH MB = 2 x FETCH_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section), WRITE MB = WRITE_SIZE (uncalibrated); "
              "per launch.  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES as reported (relative between kernels only).\n\n"
              "| kernel (workgroups) | LDS conflict / active | wait_inst | wait_any | active | VALU / MFMA insts | L2 hit | FETCH MB | WRITE MB | MFMA busy |\n"
              "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|\n")
    if agg:
        out.write(f"**gemm_nt_kernel<*> aggregate: {agg['launches']} launches, FETCH {agg['fetch_mb_per_launch']:.1f} MB and WRITE "
                  f"{agg['write_mb_per_launch']:.1f} MB per launch**\n\n")
    for (k, grid), *v in rows:
        out.write(f"| {k} grid={grid} | {v[0]:.3f} | {v[1]:.2f} | {v[2]:.2f} | {v[3]:.2f} | {v[4]:.2f} | {v[5]:.2f} | {v[6]:.0f} | {v[7]:.0f} | {v[8]:.2f} |\n")
print(open(dst).read())
