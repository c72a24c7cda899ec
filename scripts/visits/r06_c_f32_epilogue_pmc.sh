#!/bin/bash
# round 6, visit c: (i) same-box A/B, f32 streams: base (attention swizzle + bf16 dy, f32 residual loaded strip by strip) against
# new (f32 residual in batches: pairs of strips in the 8-wave kernel, one batch in the one-wave-per-SIMD kernel); (ii) kernel table of
# new; (iii) phase stamps of the LM attention kernels; (iv) in-situ PMC passes in the f32 mode -> pmc_gemm_traffic_b32_mlp_f32.json
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r06_c; mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests -q -m gpu -x -k "gemm or full_depth or recipe or b32_step or three_training" > $O/pytest_subset.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_subset.log
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-logits-full"
out=$O/ab_f32.txt; : > $out
for i in 1 2 3; do
  for lib in base new; do
    if [ $lib = new ]; then unset TA355_LIB; else export TA355_LIB=$PWD/tiny_audio_amd/libta355_$lib.so; fi
    echo -n "f32 $lib run $i: " >> $out
    timeout 200 $B 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['gemm_ms_per_step'], d['final_loss'])" >> $out 2>&1
  done
done
unset TA355_LIB
cat $out
OUT=/tmp/prof_new
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline > $OUT.log 2>&1 < /dev/null)
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python scripts/summarize_trace_steps.py "$T" $O/kernel_steps_new_f32.md --skip 2 --note "bench.py (f32 streams) under rocprofv3 --kernel-trace, library = new" | tail -1
head -30 $O/kernel_steps_new_f32.md
timeout 300 python scripts/attn_stamps.py --streams f32 --out $O/attn_stamps_f32.txt 2>&1 | grep -v Warning | tail -70
PMC_CMD="python $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-logits-full" bash scripts/gpu_pmc.sh r06_c_in_situ 2>&1 | tail -6
python scripts/summarize_pmc.py gpurun_out/pmc_r06_c_in_situ $O/pmc_summary_in_situ_f32.md --json $O/pmc_gemm_traffic_b32_mlp_f32.json --note "in situ: bench.py --steps 1 --warmup 1 (B=32, MLP projector, f32 streams = the default), every GEMM / attention launch of two training steps; round 6" | head -30
rm -rf gpurun_out/pmc_r06_c_in_situ
