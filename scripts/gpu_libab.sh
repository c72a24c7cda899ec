#!/bin/bash
# per-kernel GEMM averages of several builds of the library (TA355_LIB) under rocprofv3, one visit
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in $LIBS; do
  TA355_LIB=$REPO/tiny_audio_amd/$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$lib -o b -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python $REPO/scripts/summarize_rocprof.py $REPO/gpurun_out/prof_$lib/b_kernel_stats.csv /tmp/$lib.md --steps 4 > /dev/null
  echo "== $lib"; grep -E "${PAT:-gemm_nt_kernel|steps profiled}" /tmp/$lib.md | head -9
  rm -rf $REPO/gpurun_out/prof_$lib
done
