#!/bin/bash
# generic interleaved A/B over one environment variable for config 2 and config 3: scratch/ab_env.sh OUT VAR A B
out=$1; var=$2; a=$3; b=$4
for rep in 1 2; do
for v in "$a" "$b"; do
  env $var=$v python bench.py --pre-extracted --steps 200 --warmup 20 --skip-cpu-baseline --skip-other-configs --skip-b512 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $var=$v ms_per_step', d['ms_per_step'], 'train', d['train_graph_ms'])" >> $out
  env $var=$v python bench.py --steps 200 --warmup 20 --skip-cpu-baseline --skip-other-configs --skip-b512 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 $var=$v ms_per_step', d['ms_per_step'], 'train', d['train_graph_ms'], 'stft', d['roofline']['kernel_ms']['stft_mel'])" >> $out
done
done
cat $out
