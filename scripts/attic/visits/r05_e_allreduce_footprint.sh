#!/bin/bash
# round 5 visit e: footprint of a collective next to the step (scripts/allreduce_footprint.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python scripts/allreduce_footprint.py --ms 0.3 1.0 --wgs 16 32 64 > gpurun_out/r05_e_allreduce_footprint_mlp.txt 2>&1
tail -8 gpurun_out/r05_e_allreduce_footprint_mlp.txt
timeout 500 python scripts/allreduce_footprint.py --full-ft --ms 12 --wgs 32 --steps 8 > gpurun_out/r05_e_allreduce_footprint_fullft.txt 2>&1
tail -4 gpurun_out/r05_e_allreduce_footprint_fullft.txt
