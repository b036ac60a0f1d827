#!/bin/bash
# A/B of two library builds on the stand-alone MFCC stage (scratch/time_mfcc.py: 1024 clips, whole chip): $1 = other library
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  echo "-- tree build";  python $R/scratch/time_mfcc2.py 1024 0 256 2>&1 | grep mask
  echo "-- $1";          LIPASR_LIBRARY=$R/$1 python $R/scratch/time_mfcc2.py 1024 0 256 2>&1 | grep mask
done
