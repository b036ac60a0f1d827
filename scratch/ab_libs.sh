#!/bin/bash
# config 3 (and optionally config 2) with several library builds, interleaved, two rounds: scratch/ab_libs.sh OUT lib1 lib2 ...   ("default" = the tree's library)
out=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "default" ]; then unset LIPASR_LIBRARY; else export LIPASR_LIBRARY=$GRAFT_REPO_ROOT/asr-using-robust-nn_amd/build/liblipasr_$lib.so; fi
  python bench.py --steps 200 --warmup 20 --skip-cpu-baseline --skip-other-configs --skip-b512 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg3 $lib ms_per_step', d['ms_per_step'], 'train', d['train_graph_ms'], 'loss', d['loss'])" >> $out
  if [ -n "$AB_CFG2" ]; then python bench.py --pre-extracted --steps 200 --warmup 20 --skip-cpu-baseline --skip-other-configs --skip-b512 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $lib ms_per_step', d['ms_per_step'], 'train', d['train_graph_ms'])" >> $out; fi
done
done
cat $out
