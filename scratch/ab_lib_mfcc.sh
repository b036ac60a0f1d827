#!/bin/bash
# stand-alone MFCC stage with several library builds, interleaved, two rounds: scratch/ab_lib_mfcc.sh OUT lib1 lib2 ...
out=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "default" ]; then r=$(python scratch/time_mfcc2.py 1024 0 | tail -1); else r=$(LIPASR_LIBRARY=$lib python scratch/time_mfcc2.py 1024 0 | tail -1); fi
  echo "$lib: $r" >> $out
done
done
cat $out
