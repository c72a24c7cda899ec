#!/bin/bash
# round 5 visit j: the driver's N > 1 invocation with two ranks sharing the one GPU over gloo, for every training configuration
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for cfg in "--projector moe" "--lora" "--full-ft" "--projector qformer"; do
  tag=$(echo "$cfg" | tr -d ' -')
  timeout 500 python bench.py $cfg --gpus 2 --dist-backend gloo --share-gpu --steps 4 --warmup 2 --no-cpu-baseline --no-logits-full --no-roofline < /dev/null > gpurun_out/r05_j_2ranks_$tag.json 2> gpurun_out/r05_j_2ranks_$tag.err
  echo "$cfg rc=$?"
  python -c "
import json
d=json.loads(open('gpurun_out/r05_j_2ranks_$tag.json').read().strip().splitlines()[-1])
print('  ', d.get('error'), d.get('ms_per_step'), d.get('final_loss'), (d.get('replicas') or {}).get('replicas_identical'), (d.get('replicas') or {}).get('global_step'), (d.get('allreduce') or {}).get('bytes'))"
done
