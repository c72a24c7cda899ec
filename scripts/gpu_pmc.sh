#!/bin/bash
# PMC counter passes (separate runs, --kernel-trace only: no sys/hip trace) on a command; CSVs -> gpurun_out/pmc_<tag>/
# Rows of kernels the summary does not use (torch's init kernels, ...) are dropped on the box: the merge back is capped at 64 MiB.
TAG=${1:-r1}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD=${PMC_CMD:-"python $REPO/scripts/gemm_bench.py --reps 3"}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o pass$i -- $CMD > $OUT/pass$i.log 2>&1
  echo "pass $i ($set) rc=$?"
  rm -f $OUT/pass${i}_kernel_trace.csv
  python - "$OUT/pass${i}_counter_collection.csv" <<'PY'
import csv, sys
p = sys.argv[1]
keep = ("gemm_nt", "attn_", "lora_", "linear_small", "dec_", "logmel_fft")
with open(p) as f:
    r = csv.reader(f); hdr = next(r); k = hdr.index("Kernel_Name")
    rows = [row for row in r if any(s in row[k] for s in keep)]
with open(p, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_ALL); w.writerow(hdr); w.writerows(rows)
print(p, len(rows), "rows kept")
PY
done
du -sh $OUT
