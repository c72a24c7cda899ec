#!/bin/bash
# round 4, visit zg: implicit host<->device synchronisations inside a MoE / MLP step (torch.cuda.set_sync_debug_mode)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04_zg
for cfg in "--projector moe" ""; do
python - $cfg <<'PY' 2>&1 | grep -v "^$" | grep -i -B1 -A6 "synchroniz" | head -60
import sys, runpy, warnings, torch
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode(1)
sys.argv = ["bench.py"] + sys.argv[1:] + ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-logits-full", "--no-roofline"]
runpy.run_path("bench.py", run_name="__main__")
PY
echo "=== end cfg [$cfg]"
done > gpurun_out/r04_zg/sync_debug.txt 2>&1
wc -l gpurun_out/r04_zg/sync_debug.txt; head -80 gpurun_out/r04_zg/sync_debug.txt
