#!/bin/bash
# round 6, visit b: (i) GPU suite on the new library; (ii) same-box A/B of the step, both stream modes, three libraries:
# prev = visit a's, swz = + transposed-read swizzle of the LM attention tiles, new = + f32 residual as accumulator start + bf16 dy in the
# fp32-stream backward; (iii) per-step kernel table of `new` (f32 streams); (iv) phase stamps of the LM attention kernels
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r06_b; mkdir -p $O
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests -q -m gpu -rs -x > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
B="python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-logits-full"
for mode in f32 bf16; do
  out=$O/ab_$mode.txt; : > $out
  for i in 1 2 3; do
    for lib in prev swz new; do
      if [ $lib = new ]; then unset TA355_LIB; else export TA355_LIB=$PWD/tiny_audio_amd/libta355_$lib.so; fi
      echo -n "$mode $lib run $i: " >> $out
      timeout 200 $B --streams $mode 2>/dev/null < /dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']; print(d['ms_per_step'], d['value'], r['frac'], r['gemm_ms_per_step'], d['final_loss'])" >> $out 2>&1
    done
  done
  unset TA355_LIB
  cat $out
done
OUT=/tmp/prof_new
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o b -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline > $OUT.log 2>&1 < /dev/null)
T=$(find $OUT -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python scripts/summarize_trace_steps.py "$T" $O/kernel_steps_new_f32.md --skip 2 --note "bench.py (f32 streams) under rocprofv3 --kernel-trace, library = new" | tail -1
grep -E "attn_|rmsnorm_bwd|v4<320, 0, false|gemm_nt_kernel_v5" $O/kernel_steps_new_f32.md
timeout 300 python scripts/attn_stamps.py --streams f32 --out $O/attn_stamps_f32.txt 2>&1 | tail -60
