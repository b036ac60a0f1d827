#!/bin/bash
# build liblipasr from a git revision into scratch/liblipasr_<tag>.so (A/B timing on one GPU box)
set -e
REV=$1; TAG=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$ROOT/build/rev_$TAG
rm -rf "$W"; mkdir -p "$W/include" "$W/p/csrc"
git -C "$ROOT" show $REV:include/lipasr.h > "$W/include/lipasr.h"
for f in $(git -C "$ROOT" ls-tree --name-only $REV asr-using-robust-nn_amd/csrc/); do git -C "$ROOT" show $REV:$f > "$W/p/csrc/$(basename $f)"; done
cd "$W/p"
for f in csrc/*.hip; do /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -c $f -o $(basename $f .hip).o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o "$ROOT/scratch/liblipasr_$TAG.so"
rm -rf "$W"
ls -la "$ROOT/scratch/liblipasr_$TAG.so"
