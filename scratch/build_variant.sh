#!/bin/bash
# build/liblipasr_<tag>.so = the current library with ONE source recompiled with extra flags (A/B timing through LIPASR_LIBRARY).
# usage: scratch/build_variant.sh <tag> <source.hip> <extra flags ...>
set -e
tag=$1; src=$2; shift 2
P=asr-using-robust-nn_amd
base="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -Wall -Wno-unused-function -fno-gpu-rdc"
extra=""
[ "$src" = "stft_bdft.hip" ] && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc $base $extra "$@" -c $P/csrc/$src -o $P/build/${src%.hip}_$tag.o
objs=""
for s in core spectral dense optim mfcc stft_bdft; do
  if [ "$s.hip" = "$src" ]; then objs="$objs $P/build/${s}_$tag.o"; else objs="$objs $P/build/$s.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $P/build/liblipasr_$tag.so
echo $P/build/liblipasr_$tag.so
